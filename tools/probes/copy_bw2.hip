// copy_bw2.hip -- PROBE (tools only): what do HBM-bound streams reach on this box, against the guide's figures
// (MI355X_MICROARCH.md: "6.29 TB/s measured (float4 copy)", "1.2 GB table swept in order 6.0-6.1 TB/s")?  VERDICT r3 next #8.
// Variants beyond round 2's tools/probes/copy_bw.hip (best there: 5.37 TB/s, chunk-per-workgroup copy of 2 GiB):
//   read-only sweep (sum), write-only fill, copy; grid-stride vs chunk per workgroup vs chunk per XCD (workgroups with equal
//   id mod 8 share an XCD: each XCD walks its own contiguous eighth); 256 / 512 / 1024 threads; unroll 4 / 8 / 16; nt loads and
//   stores; buffers of 1, 2 and 4 GiB (all far beyond the 256 MiB Infinity Cache).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/copy_bw2 tools/probes/copy_bw2.hip ; run: ./tools/probes/copy_bw2
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ f4 ld(const f4 *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st(f4 *p, f4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// MODE 0 = copy, 1 = read-only (sum into a sink nobody reads unless the sum is NaN), 2 = write-only.  MAP 0 = grid-stride,
// 1 = contiguous chunk per workgroup, 2 = contiguous eighth per XCD (blockIdx % 8), workgroups of an XCD interleaved inside it.
template <int MODE, int MAP, int U, int T, bool NTL, bool NTS>
__global__ __launch_bounds__(T) void stream_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n, float *sink) {
  size_t i0, step, end;
  if (MAP == 0) { i0 = (size_t)blockIdx.x * T + threadIdx.x; step = (size_t)gridDim.x * T; end = n; }
  else if (MAP == 1) { const size_t per = n / gridDim.x; i0 = (size_t)blockIdx.x * per + threadIdx.x; step = T; end = (size_t)(blockIdx.x + 1) * per; }
  else {
    const size_t per = n / 8; const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    i0 = (size_t)x * per + (size_t)slot * T + threadIdx.x; step = (size_t)nslot * T; end = (size_t)(x + 1) * per;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = i0; i + (U - 1) * step < end; i += U * step) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = MODE == 2 ? (f4){1.f, 2.f, 3.f, (float)u} : ld<NTL>(src + i + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 1) acc += v[u];
      else st<NTS>(dst + i + u * step, v[u]);
    }
  }
  if (MODE == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <typename F>
static double run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 6;
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps * 1e-3;
}

#define RUN(MODE, MAP, U, T, NTL, NTS, blocks, label)                                                                          \
  do {                                                                                                                           \
    const double t = run([&] { hipLaunchKernelGGL((stream_kernel<MODE, MAP, U, T, NTL, NTS>), dim3(blocks), dim3(T), 0, 0, src, dst, n, sink); }); \
    const double moved = (MODE == 0 ? 2.0 : 1.0) * bytes;                                                                        \
    printf("%-10s %-26s U=%-2d T=%-4d nt(l,s)=%d%d blocks=%-6d : %6.2f TB/s\n", MODE == 0 ? "copy" : MODE == 1 ? "read" : "write", label, U, T, NTL, NTS, \
           (int)(blocks), moved / t / 1e12);                                                                                     \
    fflush(stdout);                                                                                                              \
  } while (0)

int main() {
  float *sink;
  hipMalloc(&sink, 64);
  for (size_t gib : {1, 2, 4}) {
    const size_t bytes = gib << 30, n = bytes / 16;
    f4 *src, *dst;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess) { printf("alloc of 2 x %zu GiB failed\n", gib); return 1; }
    hipMemset(src, 1, bytes); hipMemset(dst, 0, bytes);
    printf("---- buffers of %zu GiB (copy moves 2x that) ----\n", gib);
    // round 2's best shape, then the variations
    RUN(0, 1, 4, 256, false, false, 256 * 32, "chunk/workgroup");
    RUN(0, 1, 8, 256, false, false, 256 * 32, "chunk/workgroup");
    RUN(0, 1, 4, 256, true, true, 256 * 32, "chunk/workgroup");
    RUN(0, 1, 4, 256, false, true, 256 * 32, "chunk/workgroup");
    RUN(0, 1, 4, 1024, false, false, 256 * 8, "chunk/workgroup");
    RUN(0, 1, 16, 256, false, false, 256 * 16, "chunk/workgroup");
    RUN(0, 2, 4, 256, false, false, 256 * 32, "eighth/XCD");
    RUN(0, 2, 8, 512, false, false, 256 * 8, "eighth/XCD");
    RUN(0, 2, 4, 256, true, true, 256 * 32, "eighth/XCD");
    RUN(0, 0, 4, 256, false, false, 256 * 32, "grid-stride");
    RUN(0, 0, 4, 256, true, true, 256 * 32, "grid-stride");
    RUN(1, 1, 4, 256, false, false, 256 * 32, "chunk/workgroup");
    RUN(1, 1, 8, 256, false, false, 256 * 32, "chunk/workgroup");
    RUN(1, 1, 8, 256, true, false, 256 * 32, "chunk/workgroup");
    RUN(1, 2, 8, 256, false, false, 256 * 32, "eighth/XCD");
    RUN(1, 0, 8, 256, false, false, 256 * 32, "grid-stride");
    RUN(2, 1, 4, 256, false, false, 256 * 32, "chunk/workgroup");
    RUN(2, 1, 4, 256, false, true, 256 * 32, "chunk/workgroup");
    RUN(2, 2, 4, 256, false, false, 256 * 32, "eighth/XCD");
    {
      const double t = run([&] { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); });
      printf("%-10s hipMemcpyDtoD                                                    : %6.2f TB/s\n", "copy", 2.0 * bytes / t / 1e12);
    }
    hipFree(src); hipFree(dst);
  }
  return 0;
}
