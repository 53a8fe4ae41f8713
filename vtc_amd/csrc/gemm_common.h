// gemm_common.h -- MFMA wrappers, LDS-DMA staging and parameter block of the GEMM kernel (gemm.hip).
#pragma once
#include "common.h"

#include <type_traits>

namespace vtcgemm {

constexpr int ROWB = 128;          // bytes of K per LDS row
constexpr int SUPER_ROWS = 1024;   // output rows per L2 super-row

template <typename T> struct Mma;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <> struct Mma<bf16_t> {
  static constexpr int KPR = 64;  // K elements per 128-byte row
  __device__ static __forceinline__ void run(const u32x4v &w, const u32x4v &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc,
                                                  0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  static constexpr int KPR = 64;
  __device__ static __forceinline__ void run(const u32x4v &w, const u32x4v &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KPR = 32;
  __device__ static __forceinline__ void run(const u32x4v &w, const u32x4v &a, f32x4 &acc) {
    const f32x4 wv = __builtin_bit_cast(f32x4, w), av = __builtin_bit_cast(f32x4, a);   // whole-tuple casts
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], av[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], av[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], av[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], av[3], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.y), __builtin_bit_cast(float, a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.z), __builtin_bit_cast(float, a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.w), __builtin_bit_cast(float, a.w), acc, 0, 0, 0);
  }
};

struct GemmParams {
  const char *A;
  const char *W;
  const float *bias;
  void *out;
  int M, N, K;
  int lda_bytes, ldw_bytes, ldo;
  int MT, NT;
  int col_group;   // phased kernel: column tiles per group of the tile walk (0 = all NT columns in one group)
  int super_tiles; // phased kernel: row tiles per super-row of the tile walk (0 = SUPER_ROWS / 256)
  int stagger_groups, stagger_ticks;   // phased kernel: workgroup (slot % groups) starts (slot % groups) * ticks x 10 ns late
#if defined(VTC_GEMM_STAMPS) || defined(VTC_GEMM_PHASE_STAMPS)
  unsigned long long *dbg;   // diagnostic build: per-wave phase cycle sums
#endif
  GemmEpi epi;
};

typedef __attribute__((address_space(3))) void lds_void;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB).
// Inline asm so that hipcc does NOT model it as a memory operation: with the builtin the compiler
// waits vmcnt(0) before the first ds_read of every K-step (it cannot prove the DMA into the other
// buffer does not alias the reads) and the pipeline serialises.  The completion wait is ours: one
// s_waitcnt vmcnt(0) before the barrier that publishes the buffer.  M0 carries the wave-uniform LDS
// byte address and is saved/restored inside the same statement (guide 5.7).
__device__ __forceinline__ void glds16(const char *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// 16-byte LDS read the compiler does not schedule or count: `addr` is the 32-bit LDS byte address,
// OFF an immediate (< 65536).  The data is valid only after a matching lgkm_wait (LDS reads of one
// wave return in order; every wait names the registers it guards so no consumer can move above it).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // one 128-bit register tuple (asm "v" operand)
__device__ __forceinline__ void lds_read16(u32x4 &dst, unsigned addr, int off) {
  // "memory": keeps the read below the barrier that publishes the buffer and above the one that recycles it
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait(u32x4 &x) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N));
}
template <int N, int TN>
__device__ __forceinline__ void lgkm_wait_frags(u32x4 &x, u32x4 (&w)[TN]) {
  static_assert(TN == 4 || TN == 8, "four or eight weight fragments per wave");
  if constexpr (TN == 4) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N));
  } else {
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(x), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
                 : "n"(N));
  }
}

// all twelve fragments of the phased kernel's register subtile (whatever subset was just re-read) have landed
__device__ __forceinline__ void lgkm_wait_subtile(u32x4 (&a)[4][2], u32x4 (&w)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]),
                 "+v"(a[3][1]), "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
}

// four weight fragments (one 32-column half x two K halves) have landed
__device__ __forceinline__ void lgkm_wait_w4(u32x4 (&w)[2][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[1][0]), "+v"(w[1][1]));
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// Same DMA with a wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the per-lane part is
// computed once per tile, the base advances by one scalar add per K-step -- no 64-bit VALU address
// arithmetic in the K loop (each piece's issue cost is what the partner wave's MFMAs have to cover).
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

// per-lane byte offsets of this wave's GROUPS pieces of a tile whose first row is row0 (rows clamped)
template <int GROUPS>
__device__ __forceinline__ void tile_offsets(unsigned (&off)[GROUPS], int row0, int nrows, int ld_bytes, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < GROUPS; ++q) {
    const int r = (wave * GROUPS + q) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;
    off[q] = (unsigned)gr * (unsigned)ld_bytes + c * 16;
  }
}
template <int GROUPS>
__device__ __forceinline__ void stage_tile_fast(const unsigned (&off)[GROUPS], const char *sbase, unsigned lds_tile, int wave) {
#pragma unroll
  for (int q = 0; q < GROUPS; ++q)
    glds16s(off[q], sbase, __builtin_amdgcn_readfirstlane(lds_tile + (wave * GROUPS + q) * 1024));
}

// Stage GROUPS 8-row groups of a ROWS-row x 128-byte operand tile (this wave's share).
template <int GROUPS>
__device__ __forceinline__ void stage_tile(const char *base, int row0, int nrows, int ld_bytes, int kbyte, unsigned lds_tile,
                                           int wave, int lane) {
#pragma unroll
  for (int q = 0; q < GROUPS; ++q) {
    const int group = wave * GROUPS + q;      // 8-row group
    const int r = group * 8 + (lane >> 3);    // tile row this lane fills
    const int cs = lane & 7;                  // LDS chunk slot (linear)
    const int c = cs ^ ((r >> 1) & 7);        // source chunk (swizzle on the source side)
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;         // clamp: tail rows re-read a valid row, never stored
    const char *src = base + (size_t)gr * ld_bytes + kbyte + c * 16;
    glds16(src, __builtin_amdgcn_readfirstlane(lds_tile + group * 1024));   // provably wave-uniform -> SGPR operand
  }
}

// QuickGELU x * sigmoid(1.702 x) (model/timesformer_clip_alt.py:31-33).  fp32 mode: IEEE division and
// expf; bf16 mode: v_exp_f32 + v_rcp_f32 (1 ulp each, far below the bf16 rounding of the result) --
// the IEEE division sequence alone cost ~25 % of a K = 512 tile.
template <bool ACCURATE>
__device__ __forceinline__ float quick_gelu(float x) {
  if (ACCURATE) return x / (1.0f + expf(-1.702f * x));
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x));   // 1.702 * log2(e)
}

// two values at a time in the 16-bit modes: the scale, the 1 + e and the final product as packed fp32 operations (v_pk_mul_f32 /
// v_pk_add_f32), exp2 and rcp per value -- the same operations and roundings as quick_gelu<false> on each
typedef float gelu_f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f2_t quick_gelu2(gelu_f2_t x) {
  const gelu_f2_t t = x * (gelu_f2_t){-2.4554670f, -2.4554670f};
  const gelu_f2_t d = (gelu_f2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + (gelu_f2_t){1.0f, 1.0f};
  return x * (gelu_f2_t){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

int num_cus();

}  // namespace vtcgemm
