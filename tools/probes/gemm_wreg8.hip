// gemm_wreg.hip -- PROBE (tools only; never linked into libvtc_hip.so): does the 256 x 256 bf16 GEMM main loop get faster when one
// operand skips the LDS?
//
// the product's phased kernel (8 waves, 128 x 64 wave tiles, both operands global -> LDS by LDS-DMA, fragments by
// ds_read_b128) runs at 1.06-1.15 PFLOP/s at 8192^3, 1.63 without the DMA, 1.43 without the LDS reads, 1.9 with neither (timing
// ablations of round 1): the LDS array -- one port for 192 KB of fragment reads and 64 KB of DMA writes per K-tile of 64 -- is the
// suspect.  This probe keeps the 256 x 256 tile but runs it on FOUR waves (one per SIMD, 512 registers each) with 128 x 128 wave
// tiles, the ACTIVATION operand through the LDS as before (32 KB of DMA writes + 64 KB of reads per K-tile: a third of the LDS
// traffic) and the WEIGHT operand straight from global memory into MFMA fragment registers (double-buffered, one K-tile ahead:
// 2 x 64 VGPRs).  C[M, N] (bf16) = A[M, K] W[N, K]^T, M, N multiples of 256, K a multiple of 128.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/gemm_wreg tools/probes/gemm_wreg.hip
// run:   ./tools/probes/gemm_wreg [M N K]        (prints TFLOP/s and checks sampled outputs against fp32 dot products)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>

#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 128;            // bytes of K per LDS row (64 bf16)
constexpr int NSTAGE = 4;
#ifndef ABL
#define ABL 0
#endif
constexpr int STAGE = 256 * ROWB;    // one K-tile of the activation operand: 32 KB

__device__ __forceinline__ void glds16s(unsigned voff, const char *sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
template <int OFF>
__device__ __forceinline__ void gload16(u32x4 &dst, unsigned voff, const char *sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_read16(u32x4 &dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait2(u32x4 &a, u32x4 &b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
// in-place accumulation pinned to the AGPR half: with 256 accumulator registers the builtin left hipcc no slack (it chose dst != srcC,
// 1050 v_accvgpr_mov and accumulators in scratch); dependent MFMAs are 8 apart here, the hazards the compiler cannot see are covered by hand
__device__ __forceinline__ void mfma_acc(f32x4 &acc, const u32x4 &w, const u32x4 &a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
// every weight fragment of a K-tile has landed (and everything older in the wave's in-order vmcnt queue)
template <int N>
__device__ __forceinline__ void vm_wait_w(u32x4 (&w)[4][2]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[2][0]), "+v"(w[2][1]), "+v"(w[3][0]), "+v"(w[3][1])
               : "n"(N));
}

__global__ __launch_bounds__(512, 1) void gemm_wreg8_kernel(const char *__restrict__ A, const char *__restrict__ W, unsigned short *__restrict__ C,
                                                           int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;      // 2 x 4 waves of 128 x 64
  const int g = lane >> 4, n = lane & 15;
  const int MT = M / 256, NT = N / 256, ksteps = K / 64;
  const unsigned lda = K * 2, ldw = K * 2;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  // fragment read address inside a stage: row (wr 128 + 16 i + n), 16-byte slot (4 kh + g) ^ ((n >> 1) & 7)
  const unsigned a_rd = (wr * 128 + n) * ROWB;
  const unsigned coff0 = ((0 + g) ^ ((n >> 1) & 7)) << 4, coff1 = ((4 + g) ^ ((n >> 1) & 7)) << 4;
  // weight fragment j, K-half kh of K-tile kt: 16 bytes at W[(n0 + wc 128 + 16 j + n)][kt 64 + 32 kh + 8 g ..]
  const unsigned w_vo = n * ldw + g * 16;

  // XCD-aware persistent walk (the product kernel's): workgroups with equal (id mod 8) share an XCD and walk one contiguous range of
  // tiles ordered in super-rows of 4 row blocks x all column blocks, so the panels in flight stay in that XCD's 4 MiB L2
  const int ntiles = MT * NT, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  for (int li = slot; li < nt_x; li += nb_x) {
    const int logical = start_x + li;
    const int per_super = 4 * NT;
    const int sr = logical / per_super, rem = logical - sr * per_super;
    const int gsz = min(4, MT - sr * 4);
    const int ntc = rem / gsz;
    const int m0 = (sr * 4 + (rem - ntc * gsz)) * 256, n0 = ntc * 256;
    unsigned a_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (wave * 4 + q) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      a_off[q] = (unsigned)(m0 + r) * lda + c * 16;
    }
    const char *wbase = W + (size_t)(n0 + wc * 64) * ldw;
    auto stage_a = [&](int kt, int slot) __attribute__((always_inline)) {
      const unsigned dst = lds_base + (slot % NSTAGE) * STAGE + wave * 4 * 1024;
#pragma unroll
      for (int q = 0; q < 4; ++q) glds16s(a_off[q], A + (size_t)kt * ROWB, __builtin_amdgcn_readfirstlane(dst + q * 1024));
    };
    auto load_w = [&](u32x4 (&w)[4][2], int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const char *sb = wbase + (size_t)j * 16 * ldw + (size_t)kt * ROWB;
        gload16<0>(w[j][0], w_vo, sb);
        gload16<64>(w[j][1], w_vo, sb);
      }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 w0[4][2], w1[4][2];
    // prologue: activation K-tiles 0..2 and weight K-tile 0.  TWO weight sets only: a third makes hipcc spill, and a spill of a register
    // whose global_load is still in flight (the compiler cannot see that) stores garbage
    stage_a(0, 0);
    stage_a(min(1, ksteps - 1), 1);
    stage_a(min(2, ksteps - 1), 2);
    load_w(w0, 0);
    vm_wait_w<0>(w0);
    __builtin_amdgcn_s_barrier();

    auto ktile = [&](u32x4 (&wcur)[4][2], u32x4 (&wnxt)[4][2], int t) __attribute__((always_inline)) {
      // branch-free: past the end of K the last K-tile is fetched again (into a register set / a stage nobody reads any more) --
      // conditional asm loads made hipcc spill the whole fragment set around the branches (399 registers)
#if ABL != 1 && ABL != 4      // timing ablations (WRONG results): 1 no weight loads, 2 no activation DMA, 3 no LDS reads after the first, 4 = 1 + 2, 5 no barrier
      load_w(wnxt, min(t + 1, ksteps - 1));
#endif
#if ABL != 2 && ABL != 4
      stage_a(min(t + 3, ksteps - 1), t + 3);
#endif
      const unsigned st = lds_base + (t % NSTAGE) * STAGE + a_rd;
      u32x4 a0[2], a1[2];
      lds_read16(a0[0], st + coff0, 0);
      lds_read16(a0[1], st + coff1, 0);
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
#if ABL == 3
        if (i == 0) {
#endif
        lds_read16(a1[0], st + coff0, (i + 1) * 16 * ROWB);
        lds_read16(a1[1], st + coff1, (i + 1) * 16 * ROWB);
#if ABL == 3
        }
#endif
        lgkm_wait2<2>(a0[0], a0[1]);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            mfma_acc(acc[i][j], wcur[j][kh], a0[kh]);
        if (i + 2 < 8 && ABL != 3) {
          lds_read16(a0[0], st + coff0, (i + 2) * 16 * ROWB);
          lds_read16(a0[1], st + coff1, (i + 2) * 16 * ROWB);
          lgkm_wait2<2>(a1[0], a1[1]);
        } else {
          lgkm_wait2<0>(a1[0], a1[1]);
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            mfma_acc(acc[i + 1][j], wcur[j][kh], a1[kh]);
      }
      // in-order queue, oldest first: W(t+1) 16 | A(t+2) 8 | W(t+2) 16 | A(t+3) 8 -- the next step needs W(t+1) and A(t+1) (older still)
      vm_wait_w<(ABL == 1 || ABL == 2 || ABL == 4) ? 0 : 4>(wnxt);      // W(t+1) 8 | A(t+3) 4
#if ABL != 5
      __builtin_amdgcn_s_barrier();
#endif
    };
    for (int t = 0; t < ksteps; t += 2) {      // K is a multiple of 128: whole pairs of K-tiles
      ktile(w0, w1, t);
      ktile(w1, w0, t + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the redundant tail pieces; MFMA results readable
    // epilogue (probe: straight from the MFMA layout, 8-byte stores): lane holds C[m0 + wr 128 + 16 i + n][n0 + wc 128 + 16 j + 4 g ..]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t row = (size_t)(m0 + wr * 128 + 16 * i + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint2 pk;
        auto cv = [](float f) -> unsigned { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); };
        pk.x = cv(acc[i][j][0]) | (cv(acc[i][j][1]) << 16);
        pk.y = cv(acc[i][j][2]) | (cv(acc[i][j][3]) << 16);
        *reinterpret_cast<uint2 *>(C + row * N + n0 + wc * 64 + 16 * j + 4 * g) = pk;
      }
    }
    __builtin_amdgcn_s_barrier();      // the stages are free for the next tile's prologue
  }
}

static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char **argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 8192, N = argc > 3 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  if (M % 256 || N % 256 || K % 128) { fprintf(stderr, "M, N multiples of 256, K of 128\n"); return 1; }
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  srand(1);
  for (auto &v : ha) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  for (auto &v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  unsigned short *dA, *dW, *dC;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dW, hw.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int shmem = NSTAGE * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_wreg8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
  const int grid = std::min(cus, (M / 256) * (N / 256));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_wreg8_kernel, dim3(grid), dim3(512), shmem, 0, (const char *)dA, (const char *)dW, dC, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  const int iters = 20;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_wreg8_kernel, dim3(grid), dim3(512), shmem, 0, (const char *)dA, (const char *)dW, dC, M, N, K);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("gemm_wreg8 %d x %d x %d: %.3f ms = %.1f TFLOP/s\n", M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  std::vector<unsigned short> hc((size_t)M * N);
  hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 256; ++s) {
    const int r = (int)((size_t)rand() * 7919 % M), c = (int)((size_t)rand() * 104729 % N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)r * K + k]) * bf2f(hw[(size_t)c * K + k]);
    const double got = bf2f(hc[(size_t)r * N + c]);
    worst = std::max(worst, std::abs(got - ref) / (1e-3 + std::abs(ref)));
  }
  printf("sampled relative error vs fp64 dot products: %.3e %s\n", worst, worst < 2e-2 ? "(ok: bf16 output rounding)" : "(WRONG)");
  return worst < 2e-2 ? 0 : 2;
}
