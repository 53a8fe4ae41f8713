// Probe: does the instruction offset of global_load_lds_dwordx4 apply to the LDS destination as well as to the
// global source?  One wave copies 1 KiB with offset:1024 and M0 = 2048; the kernel then dumps LDS[0..8191].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const char *src, unsigned *out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  unsigned *l = reinterpret_cast<unsigned *>(lds);
  for (int i = threadIdx.x; i < 2048; i += 64) l[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  const unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)"
               : : "v"(voff), "s"(src), "s"(base + 2048) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = l[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;   // word i at byte 4 i
  char *d; unsigned *o;
  hipMalloc(&d, 16384); hipMalloc(&o, 8192);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, d, o);
  std::vector<unsigned> r(2048);
  hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 2048; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("LDS words written: [%d, %d] (bytes %d..%d); first value %u (= source byte %u)\n", first, last, first * 4, last * 4 + 3,
         first >= 0 ? r[first] : 0, first >= 0 ? r[first] * 4 : 0);
  printf("expected if the offset applies to both: LDS bytes 3072..4095, source bytes 1024..2047\n");
  return 0;
}
