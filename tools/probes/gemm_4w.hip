// gemm_4w.hip -- PROBE (tools only; never linked into libvtc_hip.so): the K loop of hipBLASLt's gfx950 kernel class, written in HIP.
//
// profiles/r04_gemm_yardstick.md: hipBLASLt's `Custom_Cijk_..._MT256x256x64_MI16x16x1` (FOUR waves of 128 x 128, one per SIMD, operands
// staged through registers) runs 8192^3 at 1 635 TFLOP/s against 1 434-1 473 for the product's 8-wave LDS-DMA kernel, i.e. ~80 % against
// ~66-70 % of the MFMA pipe inside the K loop.  This probe asks whether that loop can be had in HIP source: 256 x 256 x 64 tiles, 4 waves
// (2 x 2) of 128 x 128 (256 accumulator AGPRs per lane, MFMA by inline asm with "+a"), BOTH operands global -> registers -> LDS
// (ds_write_b128 into the product's XOR-swizzled 128-byte-row image) -> ds_read_b128 fragments, two LDS stages of 64 KiB, ONE barrier per
// K-tile, everything software-pipelined by hand inside the wave (one wave per SIMD: nothing else hides latency):
//     K-tile t, K-half 0 (64 MFMA):  ds_write K-tile t+1 from the staging registers / global_load K-tile t+2 into them (alternating,
//                                    vmcnt(15) in front of every write), then the 16 fragment reads of K-half 1
//     K-tile t, K-half 1 (64 MFMA):  after 16 MFMA: lgkmcnt(0) + s_barrier (everybody's K-tile t+1 is in LDS), then the 16 fragment
//                                    reads of K-tile t+1's K-half 0 into the registers K-half 0 has just freed
// so every ds_read has >= 16 MFMA (256 cycles) and every global load more than one K-tile to land.  C[M, N] (bf16) = A[M, K] W[N, K]^T.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/gemm_4w tools/probes/gemm_4w.hip
// run:   ./tools/probes/gemm_4w [M N K]        (TFLOP/s + sampled check against fp64 dot products)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

#ifndef ABL
#define ABL 0      // timing ablations (WRONG results on purpose): 1 = no vmcnt waits, 2 = no ds_write, 3 = no global loads in the loop, 4 = no barrier, 5 = no fragment reads
#endif
constexpr int ROWB = 128;                 // bytes of K per LDS row (64 bf16)
constexpr int TILE = 256 * ROWB;          // one operand's K-tile: 32 KiB
constexpr int STAGE = 2 * TILE;           // A then W

__device__ __forceinline__ void gload16(u32x4 &dst, unsigned voff, const char *sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void lds_write16(unsigned addr, const u32x4 &v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_read16(u32x4 &dst, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory");
}
__device__ __forceinline__ void mfma_acc(f32x4 &acc, const u32x4 &w, const u32x4 &a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
// the value of x after the wait is what later statements use (ties the consumers below the wait)
template <int N> __device__ __forceinline__ void vm_wait1(u32x4 &x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
__device__ __forceinline__ void lgkm_wait16(u32x4 (&a)[8], u32x4 (&w)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(w[0]), "+v"(w[1]),
                 "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

__global__ __launch_bounds__(256, 1) void gemm_4w_kernel(const char *__restrict__ A, const char *__restrict__ W, unsigned short *__restrict__ C, int M,
                                                        int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;      // 2 x 2 waves of 128 x 128
  const int g = lane >> 4, n = lane & 15;
  const int MT = M / 256, NT = N / 256, ksteps = K / 64;
  const unsigned lda = K * 2, ldw = K * 2;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);

  // XCD-aware tile order (bijective): workgroups with equal id mod 8 share an XCD and take consecutive tiles of one column block
  const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m0 = (tile % MT) * 256, n0 = (tile / MT) * 256;
  if (tile >= MT * NT) return;

  // staging: load q of this wave = rows wave * 64 + q * 8 + (lane >> 3) of the operand tile, 16-byte chunk lane & 7;
  // LDS image: row r at r * 128, chunk c at slot c ^ ((r >> 1) & 7) (the product's swizzle: conflict-free ds_read_b128 fragments)
  unsigned a_off[8], w_off[8], st_off[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = wave * 64 + q * 8 + (lane >> 3), c = lane & 7;
    a_off[q] = (unsigned)(m0 + r) * lda + c * 16;
    w_off[q] = (unsigned)(n0 + r) * ldw + c * 16;
    st_off[q] = (unsigned)r * ROWB + ((c ^ ((r >> 1) & 7)) << 4);
  }
  // fragment reads: activation fragment i = rows wr * 128 + 16 i + n; weight fragment j = rows wc * 128 + 16 j + n; K-half kh: chunks 4 kh + g
  const unsigned a_rd = (wr * 128 + n) * ROWB, w_rd = TILE + (wc * 128 + n) * ROWB;
  const unsigned coff[2] = {(unsigned)(((0 + g) ^ ((n >> 1) & 7)) << 4), (unsigned)(((4 + g) ^ ((n >> 1) & 7)) << 4)};

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 sa[8], sw[8];                     // staging registers: one K-tile of this wave's share of both operands
  u32x4 fa[2][8], fw[2][8];               // fragments of the two K-halves
  auto load_tile = [&](int kt) __attribute__((always_inline)) {
    const char *ab = A + (size_t)kt * ROWB, *wb = W + (size_t)kt * ROWB;
#pragma unroll
    for (int q = 0; q < 8; ++q) gload16(sa[q], a_off[q], ab);
#pragma unroll
    for (int q = 0; q < 8; ++q) gload16(sw[q], w_off[q], wb);
  };
  auto read_half = [&](int kh, unsigned st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) lds_read16(fa[kh][i], st + a_rd + coff[kh], i * 16 * ROWB);
#pragma unroll
    for (int j = 0; j < 8; ++j) lds_read16(fw[kh][j], st + w_rd + coff[kh], j * 16 * ROWB);
  };

  // prologue: K-tile 0 -> stage 0, K-tile 1 into the staging registers, K-half 0 fragments of K-tile 0 requested
  load_tile(0);
#pragma unroll
  for (int q = 0; q < 8; ++q) { vm_wait1<0>(sa[q]); lds_write16(lds_base + st_off[q], sa[q]); }
#pragma unroll
  for (int q = 0; q < 8; ++q) { vm_wait1<0>(sw[q]); lds_write16(lds_base + TILE + st_off[q], sw[q]); }
  load_tile(min(1, ksteps - 1));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_half(0, lds_base);

  for (int t = 0; t < ksteps; ++t) {
    const unsigned st_cur = lds_base + (t & 1) * STAGE, st_nxt = lds_base + ((t + 1) & 1) * STAGE;
    const int k2 = min(t + 2, ksteps - 1);        // past the end: the last K-tile again (into registers / a stage nobody reads)
    const char *ab2 = A + (size_t)k2 * ROWB, *wb2 = W + (size_t)k2 * ROWB;
    lgkm_wait16(fa[0], fw[0]);                    // K-half 0 fragments (requested >= 32 MFMA ago)
    // ---- K-half 0: 64 MFMA; slots 0-31: write K-tile t+1 / reload the register with K-tile t+2; slots 32-47: K-half 1 reads ----
    static_for<64>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      constexpr int i = s >> 3, j = s & 7;
      mfma_acc(acc[i][j], fw[0][j], fa[0][i]);
      if constexpr (s < 32) {
        constexpr int q = s >> 1;                 // 0..15: A pieces 0..7, then W pieces 0..7
        if constexpr ((s & 1) == 0) {
          // in-order vmcnt queue: loads(t+1)[q..15] and loads(t+2)[0..q-1] are outstanding: 16 -- the oldest is the one written now
          if constexpr (q < 8) {
            if (ABL != 1 && ABL != 3) vm_wait1<15>(sa[q]);
            if (ABL != 2) lds_write16(st_nxt + st_off[q], sa[q]);
          } else {
            if (ABL != 1 && ABL != 3) vm_wait1<15>(sw[q - 8]);
            if (ABL != 2) lds_write16(st_nxt + TILE + st_off[q - 8], sw[q - 8]);
          }
        } else {
          if (ABL != 3) {
            if constexpr (q < 8) gload16(sa[q], a_off[q], ab2);
            else gload16(sw[q - 8], w_off[q - 8], wb2);
          }
        }
      } else if constexpr (s < 48) {
        constexpr int r = s - 32;
        if (ABL != 5) {
          if constexpr (r < 8) lds_read16(fa[1][r], st_cur + a_rd + coff[1], r * 16 * ROWB);
          else lds_read16(fw[1][r - 8], st_cur + w_rd + coff[1], (r - 8) * 16 * ROWB);
        }
      }
    });
    lgkm_wait16(fa[1], fw[1]);                    // K-half 1 fragments (>= 16 MFMA ago) -- and this wave's 16 ds_writes
    // ---- K-half 1: 64 MFMA; after slot 15: barrier (K-tile t+1 is in LDS for everybody); slots 16-31: K-tile t+1's K-half 0 reads ----
    static_for<64>([&](auto s_c) __attribute__((always_inline)) {
      constexpr int s = decltype(s_c)::value;
      constexpr int i = s >> 3, j = s & 7;
      mfma_acc(acc[i][j], fw[1][j], fa[1][i]);
      if constexpr (s == 15) { if (ABL != 4) __builtin_amdgcn_s_barrier(); }
      if constexpr (s >= 16 && s < 32) {
        constexpr int r = s - 16;
        if (ABL != 5) {
          if constexpr (r < 8) lds_read16(fa[0][r], st_nxt + a_rd + coff[0], r * 16 * ROWB);
          else lds_read16(fw[0][r - 8], st_nxt + w_rd + coff[0], (r - 8) * 16 * ROWB);
        }
      }
    });
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the redundant tail pieces; MFMA results readable
  // epilogue (probe: straight from the MFMA layout, 8-byte stores): lane holds C[m0 + wr 128 + 16 i + n][n0 + wc 128 + 16 j + 4 g ..]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t row = (size_t)(m0 + wr * 128 + 16 * i + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint2 pk;
      auto cv = [](float f) -> unsigned { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)f); };
      pk.x = cv(acc[i][j][0]) | (cv(acc[i][j][1]) << 16);
      pk.y = cv(acc[i][j][2]) | (cv(acc[i][j][3]) << 16);
      *reinterpret_cast<uint2 *>(C + row * N + n0 + wc * 128 + 16 * j + 4 * g) = pk;
    }
  }
}

static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char **argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 8192, N = argc > 3 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  if (M % 256 || N % 256 || K % 64 || K < 128) { fprintf(stderr, "M, N multiples of 256, K a multiple of 64, >= 128\n"); return 1; }
  std::vector<unsigned short> ha((size_t)M * K), hw((size_t)N * K);
  srand(1);
  for (auto &v : ha) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  for (auto &v : hw) v = f2bf((rand() / (float)RAND_MAX - 0.5f));
  unsigned short *dA, *dW, *dC;
  if (hipMalloc(&dA, ha.size() * 2) != hipSuccess || hipMalloc(&dW, hw.size() * 2) != hipSuccess || hipMalloc(&dC, (size_t)M * N * 2) != hipSuccess) return 1;
  (void)hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(dW, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  const int shmem = 2 * STAGE;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
  const int grid = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_4w_kernel, dim3(grid), dim3(256), shmem, 0, (const char *)dA, (const char *)dW, dC, M, N, K);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  const int iters = 20;
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_4w_kernel, dim3(grid), dim3(256), shmem, 0, (const char *)dA, (const char *)dW, dC, M, N, K);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("gemm_4w %d x %d x %d: %.3f ms = %.1f TFLOP/s\n", M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  std::vector<unsigned short> hc((size_t)M * N);
  (void)hipMemcpy(hc.data(), dC, hc.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 512; ++s) {
    const int r = (int)((size_t)rand() * 7919 % M), c = (int)((size_t)rand() * 104729 % N);
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)r * K + k]) * bf2f(hw[(size_t)c * K + k]);
    const double got = bf2f(hc[(size_t)r * N + c]);
    worst = std::max(worst, std::abs(got - ref) / (1e-3 + std::abs(ref)));
  }
  printf("sampled relative error vs fp64 dot products: %.3e %s\n", worst, worst < 2e-2 ? "(ok: bf16 output rounding)" : "(WRONG)");
  return worst < 2e-2 ? 0 : 2;
}
