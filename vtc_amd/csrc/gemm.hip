// gemm.hip -- the one GEMM of the hot path:  out[M,N] = epi(A[M,K] @ W[N,K]^T + bias)
//
// Used for every projection of the towers (QKV, out-proj, temporal_fc, c_fc, c_proj, patch
// embedding, output projections, the CAM blocks), for the batch similarity and for the N x N
// distance matrix of the retrieval sweep.  Replaces the cuBLAS/MKL GEMMs behind
// model/timesformer_clip_alt.py:50,65,148,174 and upstream nn.MultiheadAttention / nn.Linear.
//
// gfx950 design
//   * two tile configurations of one kernel template:
//       "big"   256x256 output tile, 512 threads (8 waves as 2x4, 128x64 per wave), one workgroup
//               per CU, 128 KiB LDS -- 131 FLOP per byte staged from L2, used when the problem
//               has enough tiles to fill the chip;
//       "small" 128x128 tile, 256 threads (4 waves as 2x2, 64x64 per wave), two workgroups per CU
//               -- for the skinny problems (CAM, output projections, edge cases) and for fp32;
//   * K is consumed in 128-byte rows (64 bf16 / 32 fp32 per step); both operands go L2 -> LDS with
//     global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), double buffered, issued from inline
//     asm so the compiler does not serialise the pipeline behind them; one s_waitcnt vmcnt(0) +
//     barrier per K-step publishes the next buffer;
//   * LDS image is lane-linear (a DMA constraint); the XOR bank swizzle is applied on the per-lane
//     SOURCE address and again on the ds_read_b128 address (guide rule 21): conflict-free reads;
//   * bf16: v_mfma_f32_16x16x32_bf16; fp32: 4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk (exact
//     fp32; the k order inside a chunk is permuted identically for both operands);
//   * the MFMA "A" operand is the WEIGHT fragment and "B" the ACTIVATION fragment, so a lane ends
//     up holding 4 consecutive output columns of one output row; interior tiles (every tile of the
//     towers) transpose their outputs through the LDS stage the last K-step freed and store whole
//     128 / 256-byte row segments (tile_epilogue: bias / folded LayerNorm / QuickGELU / residual on
//     the (hi, lo) stream / scatter / distance applied on the way), edge tiles store straight from
//     the accumulators; a tile's small vectors are requested back to back before their first use;
//   * PERSISTENT workgroups: the grid is one (big) or two (small) workgroups per CU; each walks a
//     strided list of tiles, and the first K-slab of the NEXT tile is already in flight during the
//     last K-step of the current one, so short-K problems (K = 512/768 here) do not pay a cold
//     prologue per tile;
//   * workgroup -> tile map is XCD-aware: workgroups with equal (id mod 8) share an XCD (observed
//     dispatch; used for speed only) and walk a contiguous range of tiles ordered in super-rows of
//     1024 output rows x all columns, so the weight panels and the activation panels in flight stay
//     resident in that XCD's 4 MiB L2.
#include "gemm_common.h"
#include "ln_row.h"

// Production source: the ablation / timing-probe branches of rounds 1-2 (builds that gave WRONG results on purpose: stores,
// LDS reads, DMA or waits removed, 32x32x16 MFMA probe) are gone from this file -- their measurements are in DESIGN.md 4.1 and
// the code in the history (commit 3ac3430).  What remains are the cycle-stamp diagnostics (correct results, slower), kept out of
// line in gemm_stamps.h and compiled only with -DVTC_GEMM_STAMPS / -DVTC_GEMM_PHASE_STAMPS.  The three alternative K-loop schedules
// of round 4 (DEEP 2 / 3 / 4: bit-identical results, measured level or slower -- profiles/r04_experiments.txt 1, 14, 16) left the
// product source in round 5 (commit 7fd3e13 has them); __graft_entry__.build() refuses every -DVTC_* flag.
#if defined(VTC_ABLATE_STORES) || defined(VTC_ABLATE_DMA) || defined(VTC_ABLATE_HALF_DMA) || defined(VTC_ABLATE_DMA_EXEC1) || \
    defined(VTC_ABLATE_VMWAIT) || defined(VTC_ABLATE_LDSREAD) || defined(VTC_PROBE_MFMA32) || defined(VTC_PHASED_WAIT_FIRST) || \
    defined(VTC_PHASED_ONE_BARRIER) || defined(VTC_NO_RELAXED_FIRST) || defined(VTC_ROW_PANEL_PROBE)
#error "timing-probe macros are not part of the product source any more (see the note above)"
#endif
#include "gemm_stamps.h"

#ifndef VTC_MFMA_PRIO
#define VTC_MFMA_PRIO 3      // s_setprio of a wave inside its MFMA cluster.  Round 5 A/B (tools/gemm_ab.py, 6 rounds x 30 reps, bit-identical): 3 against
                             // round 4's 1: c_proj +1.2 %, QKV +1.0 %, c_fc / out-proj / text shapes +0.2 ... +0.8 % (profiles/r05_experiments.txt 5)
#endif
#ifndef VTC_GEMM_DEEP_DEFAULT
#define VTC_GEMM_DEEP_DEFAULT 1     // the 256 x 256 kernel's LDS-DMA pipeline: 1 = deep (round 4), 0 = one quarter in flight (rounds 1-3)
#endif

using namespace vtcgemm;

namespace {

// x of the lane the DPP control selects (row-local permutations: quad_perm, row_half_mirror 0x141, row_mirror 0x140)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// 16-byte global store, optionally non-temporal (streamed past L2: the store is acknowledged sooner, and the
// in-order vmcnt queue of the next tile's first K-tiles drains sooner behind it)
template <bool NT>
__device__ __forceinline__ void store16(void *o, uint4 v) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  v4u_t vv = {v.x, v.y, v.z, v.w};
  // (a run-time flag does not work: the two stores are merged and the hint is dropped)
  if constexpr (NT) __builtin_nontemporal_store(vv, reinterpret_cast<v4u_t *>(o));
  else *reinterpret_cast<v4u_t *>(o) = vv;
}
template <bool NT>
__device__ __forceinline__ void store16(void *o, float4 v) {
  store16<NT>(o, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}

// The number of 16-byte stores a wave issues LAST in tile_epilogue's interior fast paths (bf16 / f16 outputs: TM x 2, fp32 outputs:
// TM x 4, one per (fragment row, k) of the pass loops -- the static_asserts inside the paths hold them to it): gemm_phased_kernel's
// relaxed first-K-tile waits leave exactly this many vector-memory operations in flight (one definition, not a restated
// constant).  A fast path that issued FEWER trailing stores would make those waits under-wait and race the LDS-DMA.
template <typename OutT, int TM>
constexpr int fast_epilogue_trailing_stores() { return TM * (sizeof(OutT) == 2 ? 2 : 4); }

// ---- epilogue (shared by both kernels).  CONTRACT with gemm_phased_kernel's relaxed first-K-tile waits: on an interior
// tile a wave issues AT LEAST TM * 2 (bf16 out) or TM * 4 (fp32 out) vector-memory operations here (its 16-byte stores;
// the residual mode's x loads come on top), so the NST youngest operations before the next tile's first LDS-DMA piece
// all belong to this epilogue -- none of them is a DMA piece the K loop still has to wait for.
// The wave's TM x TN accumulator fragments -> out, through a scratch
// area of the dynamic LDS (byte offset scratch_off, >= 6 KiB per wave) that no DMA targets and nobody reads until
// the caller's next barrier.  (The area is named by OFFSET and re-based on the extern array here: handed over as
// a generic pointer, hipcc guards every LDS read behind the preceding global stores -- vmcnt waits that
// serialise the store stream; measured -20 % on the residual shapes.)
// ---- block-minima epilogue of the retrieval sweep (EPI_L2MIN): the distance matrix is never written ----------------------
// A key is a NON-NEGATIVE fp32 distance (negative rounding results clamp to zero) whose low 7 mantissa bits are replaced
// by an index inside the block: unsigned-integer order = distance order, ties by index, and min / max / med3 on the keys
// are single VALU instructions.  An empty slot is +inf (0x7F800000).
// (Integer min / max / med3 on the bit patterns: the float forms would each drag a canonicalising v_max_f32 x, x, x along.)
#define VTC_L2MIN_INF 0x7F800000u
__device__ __forceinline__ unsigned umed3(unsigned x, unsigned y, unsigned z) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
// (bits & ~mask) | (idx & mask): a key from a distance's bit pattern and an index inside the block, one VALU (v_bfi_b32)
__device__ __forceinline__ unsigned key_bfi(unsigned mask, unsigned idx, unsigned bits) {
  unsigned r;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(idx), "v"(bits));
  return r;
}
// the value the lane 16 / 32 away holds (lane ^ 16, lane ^ 32): v_permlane16_swap / v_permlane32_swap of a register with itself leave
// the partner half's value in one of the two results -- no LDS round trip (ds_bpermute) as __shfl_xor compiles to
__device__ __forceinline__ unsigned xor32_of(unsigned x, bool upper) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);     // r[0]: lanes 32-63 <- x[0-31];  r[1]: lanes 0-31 <- x[32-63]
  return upper ? r[0] : r[1];
}
__device__ __forceinline__ unsigned xor16_of(unsigned x, bool odd_row) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);     // r[0]: odd 16-lane rows <- the even row below;  r[1]: even rows <- the odd row above
  return odd_row ? r[0] : r[1];
}
// The NPL smallest keys seen, sorted (NPL = 4: three keys + the fourth as a bound; NPL = 2, round 5: the smallest + the second as a bound --
// half the vector instructions per value and half the plane bytes, for the recall-only sweep at small k, whose rank kernel sends a block
// to fp64 whenever its bound is in reach of the target's distance).
template <int NPL>
struct MinK {
  static_assert(NPL == 2 || NPL == 3 || NPL == 4, "two, three or four planes");
  unsigned v[NPL];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int q = 0; q < NPL; ++q) v[q] = VTC_L2MIN_INF;
  }
  __device__ __forceinline__ void insert(unsigned k) {       // NPL VALU: the slots shift up around k; v[NPL - 1] = NPL-th smallest of the old slots and k
    // slot q = med3(slot q-1, slot q, k): k <= v[q-1]: v[q-1];  between: k;  k >= v[q]: v[q]   (round 4: was min / max pairs).  ONE asm block, every slot
    // updated in place from the top down: left to hipcc the smallest slot's min was scheduled first into a temporary and copied back -- a v_mov per insert
    if constexpr (NPL == 2) {
      asm("v_med3_u32 %1, %0, %1, %2\n\tv_min_u32 %0, %0, %2" : "+v"(v[0]), "+v"(v[1]) : "v"(k));
    } else if constexpr (NPL == 3) {
      asm("v_med3_u32 %2, %1, %2, %3\n\tv_med3_u32 %1, %0, %1, %3\n\tv_min_u32 %0, %0, %3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]) : "v"(k));
    } else {     // (four slots as one asm block: 7 spilled registers in the phased kernel -- the compiler's own schedule stays)
#pragma unroll
      for (int q = NPL - 1; q >= 1; --q) v[q] = umed3(v[q - 1], v[q], k);
      v[0] = min(v[0], k);
    }
  }
  // the NPL smallest of two sorted tuples, sorted: reversed elementwise min (a bitonic sequence), then the exchange stages
  __device__ __forceinline__ void merge(const unsigned (&o)[NPL]) {
    if constexpr (NPL == 4) {
      const unsigned x0 = min(v[0], o[3]), x1 = min(v[1], o[2]), x2 = min(v[2], o[1]), x3 = min(v[3], o[0]);
      const unsigned y0 = min(x0, x2), y2 = max(x0, x2), y1 = min(x1, x3), y3 = max(x1, x3);
      v[0] = min(y0, y1); v[1] = max(y0, y1); v[2] = min(y2, y3); v[3] = max(y2, y3);
    } else if constexpr (NPL == 3) {     // the three smallest of the six = the elementwise min against the reversed partner; sorted by min3 / med3 / max3
      const unsigned x0 = min(v[0], o[2]), x1 = min(v[1], o[1]), x2 = min(v[2], o[0]);
      asm("v_min3_u32 %0, %1, %2, %3" : "=v"(v[0]) : "v"(x0), "v"(x1), "v"(x2));
      asm("v_max3_u32 %0, %1, %2, %3" : "=v"(v[2]) : "v"(x0), "v"(x1), "v"(x2));
      v[1] = umed3(x0, x1, x2);
    } else {
      const unsigned x0 = min(v[0], o[1]), x1 = min(v[1], o[0]);
      v[0] = min(x0, x1); v[1] = max(x0, x1);
    }
  }
};

// EPI_L2MIN2: the accumulators START at -(|q_m|^2 + |g_n|^2) / 2 (this lane's rows and columns in the MFMA layout), so that the K loop leaves
// q.g - (|q|^2 + |g|^2) / 2 = -d / 2 in them and the epilogue takes its keys straight from the accumulator bits -- |acc| with the index in the low
// mantissa bits, one v_bfi per key, where the four-plane form spends fma + add + max per value first.  |.| instead of the clamp at zero: a distance
// that rounding made negative by x <= eps reads x instead of 0, still within eps of the exact one.  The keys are HALF distances (recall_rank_kernel<2>
// halves its thresholds).  The next tile's norms are requested before a tile's epilogue and land under it.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void l2min2_half_norms(const GemmParams &p, int m0, int n0, float (&hr)[TM], float (&hc)[TN][4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WN, wc = wave % WN, g = lane >> 4, l15 = lane & 15;
  const int mbase = m0 + wr * TM * 16, nbase = n0 + wc * TN * 16;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + 16 * i + l15;
    hr[i] = m < p.M ? -0.5f * p.epi.rown[m] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nbase + 16 * j + 4 * g + e;
      hc[j][e] = n < p.N ? -0.5f * p.epi.coln[n] : 0.f;
    }
}

template <int WM, int WN, int TM, int TN, int NPLR, int NPL>      // NPLR / NPL: planes of the row / column direction
__device__ __forceinline__ void l2min_epilogue(f32x4 (&acc)[TM][TN], const GemmParams &p, int m0, int n0) {
  static_assert(TN == 4, "a wave's columns are one 64-column block");
  static_assert((NPLR == 4) == (NPL == 4), "full-distance keys go with four planes in both directions");
  using Min4 = MinK<NPL>;
  using MinR = MinK<NPLR>;
  // (the thread index made opaque once per tile, as in tile_epilogue: the sixteen key indices, the plane addresses and the lane's row /
  //  column offsets are then formed per tile instead of being hoisted out of the tile loop and held across the K loop)
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int lane = tid_ & 63, wave = tid_ >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int g = lane >> 4, l15 = lane & 15;
  const int nbase = n0 + wc * 64, mbase = m0 + wr * TM * 16;
  const bool cols_too = p.epi.colk != nullptr;
  const bool interior = nbase + 64 <= p.N && mbase + TM * 16 <= p.M;     // wave-uniform: no masking needed
  // |g|^2 (+ the offset) of the lane's 16 columns n = nbase + 16 j + 4 g + e; columns past N get no key
  float cn[TN][4];
  unsigned cvalid = 0;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = nbase + 16 * j + 4 * g + e;
      const bool v = n < p.N;
      if constexpr (NPL == 4) cn[j][e] = v ? p.epi.coln[n] : 0.f;       // (two / three planes: the norms went into the accumulators before the K loop)
      cvalid |= (unsigned)v << (4 * j + e);
    }
  Min4 col[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) col[j][e].init();
  const size_t rstride = (size_t)p.epi.nblk_c * p.M;     // elements between the planes (key 1, key 2, key 3, bound)
  const int cblk = nbase >> 6;
  // (the row loop exists twice: interior waves -- all of them but the last row / column of tiles -- run it without the two selects
  //  and the validity bit per value that edge waves need: ~350 of ~3 750 vector instructions per wave and tile)
  // (`cols_too` stays a RUN-TIME test inside the pass -- a not-taken scalar branch per value: made a compile-time choice (per tile, or
  //  per fragment row) the pass becomes straight-line code and hipcc spills 68-380 registers; round 5, profiles/r05_experiments.txt 3)
  auto row_pass = [&](int i, auto interior_c, MinR &row) __attribute__((always_inline)) {
    constexpr bool INTERIOR = decltype(interior_c)::value;
    const int m = mbase + 16 * i + l15;
    const bool mv = INTERIOR || m < p.M;
    float rn = 0.f;
    if constexpr (NPL == 4) rn = mv ? p.epi.rown[m] : 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // a distance that rounding made negative (a query against itself: |q|^2 in fp32 minus |bf16(q)|^2) clamps to zero --
        // as a signed-integer max, which needs no canonicalisation; its key then sorts first, as it must
        // (two planes: the accumulator holds -d / 2; the key's mask drops the sign with the index bits)
        unsigned bits;
        if constexpr (NPL != 4) bits = __float_as_uint(acc[i][j][e]);
        else bits = (unsigned)max(__float_as_int((rn - 2.0f * acc[i][j][e]) + cn[j][e]), 0);
        constexpr unsigned KMASK = NPL != 4 ? 0x8000007Fu : 127u;
        unsigned kr = key_bfi(KMASK, (unsigned)(16 * j + 4 * g + e), bits), kc = key_bfi(KMASK, (unsigned)(16 * i + l15), bits);
        if constexpr (!INTERIOR) {
          const bool v = mv && ((cvalid >> (4 * j + e)) & 1);
          kr = v ? kr : VTC_L2MIN_INF;
          kc = v ? kc : VTC_L2MIN_INF;
        }
        row.insert(kr);
        if (cols_too) col[j][e].insert(kc);
      }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mbase + 16 * i + l15;
    const bool mv = m < p.M;
    MinR row;
    row.init();
    if (interior) row_pass(i, std::true_type{}, row);
    else row_pass(i, std::false_type{}, row);
    // the row's 64 columns of this wave sit in the four lanes l15 + 16 g: merge (xor 16, xor 32)
    {
      const bool odd = (g & 1) != 0, up = g >= 2;
      unsigned o[NPLR];
#pragma unroll
      for (int q = 0; q < NPLR; ++q) o[q] = xor16_of(row.v[q], odd);
      row.merge(o);
#pragma unroll
      for (int q = 0; q < NPLR; ++q) o[q] = xor32_of(row.v[q], up);
      row.merge(o);
    }
    // lane g stores plane g (keys 1-3, bound): 16 consecutive rows each (64-byte segments).  A wave whose 64 columns lie
    // wholly past N owns no block (its slot would be the next plane's block 0).
    if (mv && nbase < p.N && g < NPLR) {
      unsigned mine = row.v[0];
#pragma unroll
      for (int q = 1; q < NPLR; ++q) mine = g == q ? row.v[q] : mine;
      p.epi.rowk[(size_t)g * rstride + (size_t)cblk * p.M + m] = mine;
    }
  }
  if (cols_too) {
    // A column's TM * 16 rows of this wave sit in the 16 lanes (l15) of its DPP row, and the lane holds 16 such columns (slot c = 4 j + e).
    // TRANSPOSE-REDUCE (round 5): at every step a lane keeps HALF of its slots, sends the other half to its partner and merges what
    // it receives -- 8 + 4 + 2 + 1 = 15 merges per lane and one finished column per lane at the end, where the all-reduce butterfly of
    // round 4 (every lane merging every slot at every step) did 64: ~520 of the ~3 400 vector instructions per wave and tile.
    // Step partners: l15 ^ 1, ^ 2 (quad_perm), ^ 8 (row_ror:8), ^ 4 (row_shl:4 / row_shr:4 under bank masks); the lane's final slot is
    // c = 8 b0 + 4 b1 + 2 b3 + b2 (b_k = bit k of l15).
    const size_t cstride = (size_t)p.epi.nblk_r * p.N;
    const int rblk = mbase / (TM * 16);
#define VTC_DPPU(x, ctrl) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, 0xf, 0xf, false)
    auto xor4 = [](unsigned x) -> unsigned {       // the value of lane l15 ^ 4: banks 0, 2 read four lanes up, banks 1, 3 four lanes down
      int r = __builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xf, 0x5, false);          // row_shl:4
      r = __builtin_amdgcn_update_dpp(r, (int)x, 0x114, 0xf, 0xA, false);              // row_shr:4
      return (unsigned)r;
    };
    // one step on a pair of slots: `up` lanes keep `hi` and send `lo`, the others keep `lo` and send `hi`
    auto step = [&](const Min4 &lo, const Min4 &hi, bool up, auto fetch) __attribute__((always_inline)) -> Min4 {
      Min4 k;
      unsigned got[NPL];
#pragma unroll
      for (int q = 0; q < NPL; ++q) {
        k.v[q] = up ? hi.v[q] : lo.v[q];
        got[q] = fetch(up ? lo.v[q] : hi.v[q]);
      }
      k.merge(got);
      return k;
    };
    const bool b0 = (l15 & 1) != 0, b1 = (l15 & 2) != 0, b2 = (l15 & 4) != 0, b3 = (l15 & 8) != 0;
    Min4 s8[8], s4[4], s2[2];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      s8[c] = step(col[c >> 2][c & 3], col[(c + 8) >> 2][c & 3], b0, [](unsigned x) { return VTC_DPPU(x, 0xB1); });     // quad_perm [1,0,3,2]
#pragma unroll
    for (int c = 0; c < 4; ++c) s4[c] = step(s8[c], s8[c + 4], b1, [](unsigned x) { return VTC_DPPU(x, 0x4E); });         // quad_perm [2,3,0,1]
#pragma unroll
    for (int c = 0; c < 2; ++c) s2[c] = step(s4[c], s4[c + 2], b3, [](unsigned x) { return VTC_DPPU(x, 0x128); });        // row_ror:8
    const Min4 fin = step(s2[0], s2[1], b2, xor4);
#undef VTC_DPPU
    const int cj = (b0 ? 2 : 0) + (b1 ? 1 : 0), ce = (b3 ? 2 : 0) + (b2 ? 1 : 0);
    const int n = nbase + 16 * cj + 4 * g + ce;
    // every lane stores the planes of its own column: a wave covers its 64 columns, 256 contiguous bytes per plane
    if (n < p.N && mbase < p.M) {
      unsigned *dst = p.epi.colk + (size_t)rblk * p.N + n;
#pragma unroll
      for (int q = 0; q < NPL; ++q) dst[q * cstride] = fin.v[q];
    }
  }
}

// REJOIN (the phased kernel): waves 0 .. WN-1 run one barrier ahead of the others and wait for them here -- AFTER the tile's small
// vectors (bias, folded-LayerNorm statistics, first residual rows) have been requested, BEFORE the LDS scratch is touched (it is the
// stage the lagging half reads last).  Every path executes the barrier exactly once.
template <typename T, int MODE_T, typename OutT, int WM, int WN, int TM, int TN, int SCRATCH_PER_WAVE, bool REJOIN = false>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[TM][TN], const GemmParams &p, int m0, int n0, unsigned scratch_off) {
  // EPI_*_FOLD = the base epilogue + the folded LayerNorm, compiled apart so that the plain kernels keep their registers
  constexpr bool FOLD = MODE_T >= EPI_FOLD_BASE && MODE_T <= EPI_RESID_FOLD_C;
  constexpr bool CENTER = MODE_T == EPI_RESID_FOLD_C;
  constexpr int MODE = CENTER ? VTC_EPI_RESID : (FOLD ? MODE_T - EPI_FOLD_BASE : MODE_T);
  auto rejoin = [&]() __attribute__((always_inline)) {
    if constexpr (REJOIN) {
      if ((int)(threadIdx.x >> 6) / WN == 0) __builtin_amdgcn_s_barrier();
    }
  };
  if constexpr (MODE == EPI_L2MIN || l2min_half_keys(MODE)) {
    rejoin();
    if constexpr (TN == 4) l2min_epilogue<WM, WN, TM, TN, l2min_row_planes(MODE), l2min_col_planes(MODE)>(acc, p, m0, n0);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  // The thread index is made opaque HERE, once per tile: everything the epilogue derives from it (row / column offsets, bias and
  // statistics addresses) is then recomputed per tile instead of being hoisted out of the persistent tile loop, where it sat in
  // registers across the K loop and was spilled (the reload at the top of the epilogue was one more dependent round trip per tile).
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int g = lane >> 4;
  const int ldo = p.ldo;
  const bool vec_ok = (ldo & 3) == 0;
  // Streaming (nt) stores.  A tile's stores sit in front of the next tile's first LDS-DMA waits in the wave's
  // in-order vmcnt queue, so the K loop restarts only once they are acknowledged.  The distance matrix (256 KiB of
  // fp32 per tile, read back once by the top-k pass): -27 % on the 50k x 50k distance GEMM; the towers' bf16
  // outputs: -1.6 % per config-2 step; the fp32 residual stream (re-read by the LayerNorm that follows): neutral,
  // kept on the default policy.  Mask bits: 0 distance, 1 bf16 outputs, 2 residual, 3 other fp32 (A/B builds).
#ifndef VTC_NT_MASK
#define VTC_NT_MASK 3
#endif
  constexpr bool nt_out = (VTC_NT_MASK >> (MODE == EPI_L2DIST ? 0 : (MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) ? 2 : sizeof(OutT) == 2 ? 1 : 3)) & 1;
  // Interior tiles (every tile of the towers) take the transposed fast epilogue; edge tiles the generic one.
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && vec_ok;
  // ---- epilogue: lane holds out[m][n..n+3], m = m0 + 16 (wr TM + i) + (lane & 15),
  //                n = n0 + 16 (wc TN + j) + 4 g
  float scale = 1.0f;
  if (MODE == EPI_SCALE) scale = __expf(*p.epi.scale_log);
  if constexpr (sizeof(OutT) == 2) {
    if (interior) {
      // bf16 outputs (plain store / QuickGELU): finalise in the MFMA layout, convert, and transpose the BF16 values
      // through LDS -- half the LDS bytes of the fp32 transposition below, whose ds_write_b128 stream (79 B/clk
      // per CU) was ~4k of the ~6k cycles of this epilogue.  Rows of 64 bf16 padded to 136 B: the ds_write_b64
      // of a 16-lane group land on 32 distinct banks.
      constexpr int TSB = 136;
      char *trb = lds + scratch_off + wave * SCRATCH_PER_WAVE;
      const int l15 = lane & 15;
      const int ncol0 = n0 + wc * TN * 16;
      // The tile's small vectors -- bias, row statistics, s -- are requested back to back and waited for ONCE (one uniform branch
      // around the bias loads: a select per load made hipcc wait for each of them in turn -- with the statistics and s, six
      // dependent round trips per tile before the first store, profiles/r04_experiments.txt 10), and before the re-join barrier.
      float4 b4[TN];
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < TN; ++j) b4[j] = *reinterpret_cast<const float4 *>(p.bias + ncol0 + 16 * j + 4 * g);
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // folded LayerNorm (GemmEpi::fold_stat): v = rstd_m (acc - mean_m s_n) + c_n, the row statistics in the MFMA layout
      // (row = lane & 15 of fragment row i), fetched before the first store of the tile
      // The wave's TM * 16 rows of (mean, rstd) and TN * 16 columns of s wait in the wave's LDS scratch behind the two
      // transposition buffers (32 + 16 registers less than holding them: the K loop leaves none to spare).
      [[maybe_unused]] float *fold_rs = reinterpret_cast<float *>(trb + 2 * 16 * TSB);
      [[maybe_unused]] float *fold_sc = fold_rs + 2 * TM * 16;
      constexpr int NQ = (TM * 16 + 63) / 64;
      [[maybe_unused]] float2 fst[NQ];
      [[maybe_unused]] float4 fsc = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (FOLD) {
        static_assert(2 * 16 * TSB + (2 * TM * 16 + TN * 16) * 4 <= SCRATCH_PER_WAVE, "folded LayerNorm: row statistics + s in the wave's scratch");
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = min(lane + 64 * q, TM * 16 - 1);
          fst[q] = *reinterpret_cast<const float2 *>(p.epi.fold_stat + 2 * (size_t)(m0 + wr * TM * 16 + r));
        }
        fsc = *reinterpret_cast<const float4 *>(p.epi.fold_s + ncol0 + 4 * (lane & (TN * 4 - 1)));
      }
      rejoin();
      if constexpr (FOLD) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = lane + 64 * q;
          if (TM * 16 % 64 == 0 || r < TM * 16) *reinterpret_cast<float2 *>(fold_rs + 2 * r) = fst[q];
        }
        if (lane < TN * 4) *reinterpret_cast<float4 *>(fold_sc + 4 * lane) = fsc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      // two buffers per wave, software-pipelined: fragment row i+1 is finalised and written while the read-back of
      // row i is in flight (a wave's LDS operations execute in order, so a buffer is rewritten only after the
      // reads of it have been issued)
      auto put = [&](int i) __attribute__((always_inline)) {
        char *buf = trb + (i & 1) * (16 * TSB);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v0, v1, v2, v3;
          if constexpr (FOLD) {
            const float2 st = *reinterpret_cast<const float2 *>(fold_rs + 2 * (i * 16 + l15));
            const float4 sj = *reinterpret_cast<const float4 *>(fold_sc + 16 * j + 4 * g);
            // two packed fmas per register pair, pairs as the accumulator holds them ((0,1), (2,3)): left to itself hipcc pairs
            // (0,2) / (1,3) and spends six moves per four values on it.  Same two roundings per value as the scalar form.
            typedef float f2_t __attribute__((ext_vector_type(2)));
            const f2_t nmu = {-st.x, -st.x}, rs = {st.y, st.y};
            const f2_t t01 = __builtin_elementwise_fma(nmu, (f2_t){sj.x, sj.y}, (f2_t){acc[i][j][0], acc[i][j][1]});
            const f2_t t23 = __builtin_elementwise_fma(nmu, (f2_t){sj.z, sj.w}, (f2_t){acc[i][j][2], acc[i][j][3]});
            const f2_t w01 = __builtin_elementwise_fma(rs, t01, (f2_t){b4[j].x, b4[j].y});
            const f2_t w23 = __builtin_elementwise_fma(rs, t23, (f2_t){b4[j].z, b4[j].w});
            v0 = w01.x; v1 = w01.y; v2 = w23.x; v3 = w23.y;
          } else {
            v0 = acc[i][j][0] + b4[j].x; v1 = acc[i][j][1] + b4[j].y; v2 = acc[i][j][2] + b4[j].z; v3 = acc[i][j][3] + b4[j].w;
          }
          if (MODE == VTC_EPI_GELU) {
            if constexpr (sizeof(T) == 4) {
              v0 = quick_gelu<true>(v0); v1 = quick_gelu<true>(v1); v2 = quick_gelu<true>(v2); v3 = quick_gelu<true>(v3);
            } else {
              const gelu_f2_t g01 = quick_gelu2((gelu_f2_t){v0, v1}), g23 = quick_gelu2((gelu_f2_t){v2, v3});
              v0 = g01.x; v1 = g01.y; v2 = g23.x; v3 = g23.y;
            }
          }
          uint2 pk;
          pk.x = pack16<OutT>(v0, v1);
          pk.y = pack16<OutT>(v2, v3);
          *reinterpret_cast<uint2 *>(buf + l15 * TSB + j * 32 + g * 8) = pk;
        }
      };
      static_assert(2 * 16 * TSB <= SCRATCH_PER_WAVE, "two transposition buffers per wave");
      static_assert(fast_epilogue_trailing_stores<OutT, TM>() == TM * 2, "16-bit fast path: TM fragment rows x 2 stores (the loop below)");
      char *out_base16;
      {
        const int m0s = __builtin_amdgcn_readfirstlane(m0), wrs = __builtin_amdgcn_readfirstlane(wr), ncs = __builtin_amdgcn_readfirstlane(ncol0);
        out_base16 = reinterpret_cast<char *>(reinterpret_cast<unsigned short *>(p.out) + (size_t)(m0s + wrs * TM * 16) * ldo + ncs);
      }
      put(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const char *buf = trb + (i & 1) * (16 * TSB);
        uint2 lo[2], hi[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int r = (lane >> 3) + 8 * k, c = lane & 7;
          lo[k] = *reinterpret_cast<const uint2 *>(buf + r * TSB + c * 16);
          hi[k] = *reinterpret_cast<const uint2 *>(buf + r * TSB + c * 16 + 8);
        }
        if (i + 1 < TM) put(i + 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          // wave-uniform base (this wave's first output element) + a 32-bit lane offset: the `saddr` form of the store
          char *o = out_base16 + (unsigned)(((i * 16 + (lane >> 3) + 8 * k) * ldo + (lane & 7) * 8) * 2);
          store16<nt_out>(o, make_uint4(lo[k].x, lo[k].y, hi[k].x, hi[k].y));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      return;
    }
  }
  // residual modes store through a wave-uniform descriptor of this tile's rows (base = row m0: 32-bit offsets whatever M)
  [[maybe_unused]] __amdgpu_buffer_rsrc_t out_rsrc;
  [[maybe_unused]] const int m0u = __builtin_amdgcn_readfirstlane(m0);
  if constexpr (MODE == EPI_RESID_LN || MODE == VTC_EPI_RESID) {
    const size_t rem = (size_t)(p.M - m0u) * ldo * 4;
    out_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float *>(p.out) + (size_t)m0u * ldo, 0,
                                                 (int)(rem < 0xFFFFFFF0u ? rem : 0xFFFFFFF0u), 0x00020000);
  }
  if (interior) {
    // Fast path (every tile of the towers).  The write path of a CU retires roughly one distinct
    // cache line per 5-8 cycles whatever its fill, so storing straight from the MFMA layout
    // (16 rows x 32..64 B per instruction) made the epilogue cost as much as 5 K-steps.  Instead
    // each wave transposes its outputs through the LDS stage that the last K-step has just freed
    // (the other stage already holds the next tile's first slab) and writes whole rows: every
    // store instruction covers 4 (fp32) or 8 (bf16) full 256 / 128-byte row segments.
    constexpr int TS = 68;                                   // padded row stride (floats): conflict-free b128 writes
    // A pass moves 16 rows x 64 columns: the wave's TN * 16 columns are H = TN / 4 column halves (H = 2 for the
    // 128 x 512 row-panel tiles), pass pp = (fragment row i, half hh)
    constexpr int H = TN / 4, NP = TM * H;
    static_assert(TN % 4 == 0 && (H == 1 || sizeof(OutT) == 4), "wide wave tiles: fp32 outputs only");
    static_assert(sizeof(OutT) == 2 || fast_epilogue_trailing_stores<OutT, TM>() <= NP * 4, "fp32 fast path: NP passes x 4 stores (>= the count the K loop's relaxed waits assume)");
    float *tr = reinterpret_cast<float *>(lds + scratch_off + wave * SCRATCH_PER_WAVE);
    const int l15 = lane & 15;
    const int ncol0 = n0 + wc * TN * 16;
    // column addend (bias, or |g|^2 of the distance epilogue) for the columns this lane writes back
    const float *colv = MODE == EPI_L2DIST ? p.epi.coln : p.bias;
    float cadd[H][8];
#pragma unroll
    for (int hh = 0; hh < H; ++hh)
#pragma unroll
      for (int e = 0; e < 8; ++e) cadd[hh][e] = 0.f;
    if (colv) {
      if constexpr (sizeof(OutT) == 4) {
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
          const float4 c4 = *reinterpret_cast<const float4 *>(colv + ncol0 + 64 * hh + l15 * 4);
          cadd[hh][0] = c4.x; cadd[hh][1] = c4.y; cadd[hh][2] = c4.z; cadd[hh][3] = c4.w;
        }
      } else {
        const float4 c0 = *reinterpret_cast<const float4 *>(colv + ncol0 + (lane & 7) * 8);
        const float4 c1 = *reinterpret_cast<const float4 *>(colv + ncol0 + (lane & 7) * 8 + 4);
        cadd[0][0] = c0.x; cadd[0][1] = c0.y; cadd[0][2] = c0.z; cadd[0][3] = c0.w;
        cadd[0][4] = c1.x; cadd[0][5] = c1.y; cadd[0][6] = c1.z; cadd[0][7] = c1.w;
      }
    }
    auto fin = [&](float a, float add) -> float {
      float v = a + add;
      if (MODE == VTC_EPI_GELU) v = quick_gelu<sizeof(T) == 4>(v);
      if (MODE == EPI_SCALE) v *= scale;
      return v;
    };
    // residual mode: the x rows of pass pp+1 are fetched while pass pp is transposed and stored
    auto x_ptr = [&](int pp, int k) -> float * {
      const int m = m0 + (wr * TM + pp / H) * 16 + (lane >> 4) + 4 * k;
      return reinterpret_cast<float *>(p.out) + (size_t)m * ldo + ncol0 + 64 * (pp % H) + l15 * 4;
    };
#ifndef VTC_RESID_AUX
#define VTC_RESID_AUX 16    // cache policy of the residual stores: 16 = sc1 (write-through)
#endif
#ifndef VTC_RESID_DEPTH
#define VTC_RESID_DEPTH 2
#endif
    // x rows of the next XD - 1 passes in flight.  Depth 2 (one pass ahead) is enough: these loads queue behind the
    // previous passes' stores in the in-order vmcnt queue, and the N = 512 residual GEMMs move their 605 MB at
    // 4.35 TB/s -- the copy rate of the chip -- at depth 2, 3 and 4 alike (3 = +14 VGPRs, 4 spills)
    constexpr int XD = VTC_RESID_DEPTH;
    float4 xr[XD][4];
    // folded LayerNorm: the residual row is the 16-bit pair (hi, lo) of GemmEpi::y16 / y16lo -- fetched as two 8-byte loads
    // into the same four registers the fp32 row would take: (hi.x, hi.y, lo.x, lo.y)
    constexpr bool SPLIT = FOLD && MODE == VTC_EPI_RESID && sizeof(T) == 2;
    // ... and kept CENTRED: every reader of the stream is a LayerNorm, which does not see a per-row constant, so the row's
    // mean BEFORE this update (fold_stat, left by the statistics pass behind the previous residual GEMM) is subtracted on the
    // way -- the stored rows keep |mean| << std, and the rounding of hi, relative to |x|, stays relative to the spread that
    // the LayerNorm divides by (tests/probes/fold_dc_probe.py: without this a DC offset of 10 std costs 1.5e-3).  The wave's
    // TM * 16 means wait in its LDS scratch behind the transposition buffer.
    // (fold_stat == NULL: this update does not re-centre -- the towers do it once per layer, in c_proj's epilogue)
    [[maybe_unused]] float *mean_prev = tr + 16 * TS;
    // ... and the partial statistics this epilogue produces (one (sum, squared deviations) pair per row of the wave's 64 columns) are
    // collected behind them and leave in ONE full-width store after the last pass: written where they arise they were 32 stores of
    // four lanes x 8 bytes per wave and tile inside the (hi, lo) store stream -- a third of its instructions, every one a partial line
    [[maybe_unused]] float2 *stat_buf = reinterpret_cast<float2 *>(tr + 16 * TS + TM * 16);
    // (requested here, with the first residual rows below and before the re-join barrier; written to the scratch after it)
    constexpr int NQM = (TM * 16 + 63) / 64;
    [[maybe_unused]] float mprev[NQM];
    if constexpr (SPLIT && CENTER) {
      static_assert((16 * TS + TM * 16) * 4 <= SCRATCH_PER_WAVE, "folded LayerNorm: previous means in the wave's scratch");
#pragma unroll
      for (int q = 0; q < NQM; ++q) mprev[q] = p.epi.fold_stat[2 * (size_t)(m0 + wr * TM * 16 + min(lane + 64 * q, TM * 16 - 1))];
    }
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    [[maybe_unused]] char *hi_base = nullptr, *lo_base = nullptr;
    [[maybe_unused]] unsigned split_off0 = 0, split_row4 = 0;
    if constexpr (SPLIT) {
      const int m0s = __builtin_amdgcn_readfirstlane(m0);
      const int wrs = __builtin_amdgcn_readfirstlane(wr), ncs = __builtin_amdgcn_readfirstlane(ncol0);
      const size_t row0 = (size_t)(m0s + wrs * TM * 16) * ldo + ncs;
      hi_base = reinterpret_cast<char *>(reinterpret_cast<unsigned short *>(p.epi.y16) + row0);
      lo_base = reinterpret_cast<char *>(reinterpret_cast<unsigned short *>(p.epi.y16lo) + row0);
      split_off0 = (unsigned)(((lane >> 4) * ldo + l15 * 4) * 2);
      split_row4 = (unsigned)ldo * 8u;
    }
    auto split_off = [&](int pp, int k) -> unsigned { return split_off0 + (unsigned)(4 * (pp / H) + k) * split_row4 + (unsigned)(128 * (pp % H)); };
    auto x_load = [&](int pp, int k) -> float4 {
      if constexpr (SPLIT) {
        const v2u_t hi = *reinterpret_cast<const v2u_t *>(hi_base + split_off(pp, k));
        const v2u_t lo = *reinterpret_cast<const v2u_t *>(lo_base + split_off(pp, k));
        return make_float4(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(lo.x), __uint_as_float(lo.y));
      } else {
        return *reinterpret_cast<const float4 *>(x_ptr(pp, k));
      }
    };
    // rows with m % skip_mod == 0 are left alone: ONE division per lane and tile, then a running remainder along the lane's rows
    // (the per-row `m % skip_mod` was ~18 vector instructions x 32 rows per wave and tile)
    [[maybe_unused]] unsigned skip_rem = 1, skip_step = 0, skip_m = 0;
    if constexpr (MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) {
      if (p.epi.skip_mod > 0) {
        skip_m = (unsigned)p.epi.skip_mod;
        skip_rem = (unsigned)(m0 + wr * TM * 16 + (lane >> 4)) % skip_m;
        skip_step = 4u % skip_m;
      }
    }
    [[maybe_unused]] bool live_k[4] = {true, true, true, true};
    if (MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) {
#pragma unroll
      for (int a = 0; a < XD - 1; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (a < NP) xr[a][k] = x_load(a, k);
    }
    // distance mode: |q|^2 of the lane's rows and |g|^2 of its columns are fetched once, in the MFMA layout, and the
    // distance is formed BEFORE the transposition: a load inside the pass loop sits behind the previous pass's
    // stores in the in-order vmcnt queue and would drain them every pass
    float rn[TM];
    if (MODE == EPI_L2DIST) {
#pragma unroll
      for (int i = 0; i < TM; ++i) rn[i] = p.epi.rown[m0 + (wr * TM + i) * 16 + l15];
    }
    rejoin();
    if constexpr (SPLIT && CENTER) {
#pragma unroll
      for (int q = 0; q < NQM; ++q) {
        const int r = lane + 64 * q;
        if (TM * 16 % 64 == 0 || r < TM * 16) mean_prev[r] = mprev[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const int i = pp / H, hh = pp % H;
      const int ncolh = ncol0 + 64 * hh;
      if ((MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) && pp + XD - 1 < NP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xr[(pp + XD - 1) % XD][k] = x_load(pp + XD - 1, k);
      }
      // 1. registers -> LDS: final fp32 values in [16 rows][64 cols]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 a4 = make_float4(acc[i][4 * hh + j][0], acc[i][4 * hh + j][1], acc[i][4 * hh + j][2], acc[i][4 * hh + j][3]);
        if (MODE == EPI_L2DIST)     // |q|^2 - 2 q.g here, + |g|^2 (a per-column addend like a bias) after the transposition
          a4 = make_float4(rn[i] - 2.0f * a4.x, rn[i] - 2.0f * a4.y, rn[i] - 2.0f * a4.z, rn[i] - 2.0f * a4.w);
        *reinterpret_cast<float4 *>(tr + l15 * TS + 16 * j + 4 * g) = a4;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // 2. LDS -> global, row-contiguous
      const int mrow0 = m0 + (wr * TM + i) * 16;
      if constexpr (sizeof(OutT) == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = (lane >> 4) + 4 * k, cc = l15 * 4;
          float4 v = *reinterpret_cast<const float4 *>(tr + r * TS + cc);
          const int m = mrow0 + r;
          v.x = fin(v.x, cadd[hh][0]); v.y = fin(v.y, cadd[hh][1]); v.z = fin(v.z, cadd[hh][2]); v.w = fin(v.w, cadd[hh][3]);
          size_t orow = (size_t)m;
          bool live = true;
          if constexpr (MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) {
            if (hh == 0) {                                       // row (i, k): next in this lane's walk
              live_k[k] = skip_rem != 0;
              skip_rem += skip_step;
              skip_rem = min(skip_rem, skip_rem - skip_m);       // unsigned: the wrapped difference loses unless skip_rem >= skip_m
            }
            live = live_k[k];
          }
          if (MODE == EPI_PATCH) {
            const int np = m % p.epi.P, ft = m / p.epi.P;
            const int tt = ft % p.epi.F, item = ft / p.epi.F;
            orow = p.epi.frames_major ? (size_t)item * p.epi.T + 1 + (size_t)tt * p.epi.P + np
                                      : (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
            const float4 p4 = *reinterpret_cast<const float4 *>(p.epi.pos + (size_t)(1 + np) * p.N + ncolh + cc);
            v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
            if (p.epi.temporal) {
              const float4 t4 = *reinterpret_cast<const float4 *>(p.epi.temporal + (size_t)tt * p.N + ncolh + cc);
              v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
            }
          }
          float *o = reinterpret_cast<float *>(p.out) + orow * ldo + ncolh + cc;
          if (MODE == EPI_RESID_LN || MODE == VTC_EPI_RESID) {
            // skipped rows are written back unchanged (a select, not a branch: an exec-masked store makes hipcc
            // re-wait on the x prefetch after every store, which throttles the store stream).  The stores are
            // WRITE-THROUGH (sc1) buffer stores: measured 8-10 % faster than plain global stores on the residual GEMMs of
            // the towers (tools/resid_ln_bench.py), and what makes the rows visible to the column tile that
            // completes the row block in EPI_RESID_LN without a release fence (MI355X_MICROARCH "Valid forms",
            // cdna_hip_programming Guideline 16 R1)
            float4 x = xr[pp % XD][k];
            if constexpr (SPLIT) {
              const unsigned h0 = __float_as_uint(x.x), h1 = __float_as_uint(x.y), l0 = __float_as_uint(x.z), l1 = __float_as_uint(x.w);
              float mu = 0.0f;
              if constexpr (CENTER) mu = mean_prev[i * 16 + r];
              x = make_float4((up16<T>((unsigned short)(h0 & 0xFFFFu)) - mu) + up16<T>((unsigned short)(l0 & 0xFFFFu)),
                              (up16<T>((unsigned short)(h0 >> 16)) - mu) + up16<T>((unsigned short)(l0 >> 16)),
                              (up16<T>((unsigned short)(h1 & 0xFFFFu)) - mu) + up16<T>((unsigned short)(l1 & 0xFFFFu)),
                              (up16<T>((unsigned short)(h1 >> 16)) - mu) + up16<T>((unsigned short)(l1 >> 16)));
            }
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            const float4 y = make_float4(live ? x.x + v.x : x.x, live ? x.y + v.y : x.y, live ? x.z + v.z : x.z, live ? x.w + v.w : x.w);
            if constexpr (!SPLIT) {
              const v4u_t yy = {__float_as_uint(y.x), __float_as_uint(y.y), __float_as_uint(y.z), __float_as_uint(y.w)};
              __builtin_amdgcn_raw_buffer_store_b128(yy, out_rsrc, (int)((((size_t)m - m0u) * ldo + ncolh + cc) * 4), 0,
                                                     MODE == EPI_RESID_LN ? 16 : VTC_RESID_AUX);
            } else {
              // producer side of the folded LayerNorm: the updated row as (hi, lo) + this wave's partial statistics
              uint2 pk, pl;
              pk.x = pack16<T>(y.x, y.y);
              pk.y = pack16<T>(y.z, y.w);
              pl.x = pack16<T>(y.x - up16<T>((unsigned short)(pk.x & 0xFFFFu)), y.y - up16<T>((unsigned short)(pk.x >> 16)));
              pl.y = pack16<T>(y.z - up16<T>((unsigned short)(pk.y & 0xFFFFu)), y.w - up16<T>((unsigned short)(pk.y >> 16)));
              // cache policy of the (hi, lo) stores: the wave's next x loads sit behind them in the in-order vmcnt queue, so how
              // soon a store is ACKNOWLEDGED sets the pace of the pass loop (0 plain, 1 nt; measured: DESIGN 4.1)
#ifndef VTC_SPLIT_ST
#define VTC_SPLIT_ST 0
#endif
              v2u_t *ph = reinterpret_cast<v2u_t *>(hi_base + split_off(pp, k));
              v2u_t *pq = reinterpret_cast<v2u_t *>(lo_base + split_off(pp, k));
              if constexpr (VTC_SPLIT_ST == 1) {
                __builtin_nontemporal_store((v2u_t){pk.x, pk.y}, ph);
                __builtin_nontemporal_store((v2u_t){pl.x, pl.y}, pq);
              } else {
                *ph = (v2u_t){pk.x, pk.y};
                *pq = (v2u_t){pl.x, pl.y};
              }
              // (sum, sum of squared deviations from the 64-column mean): merged exactly like a two-pass variance
              // (fold_stats_kernel).  Over the 16 lanes of the row: xor 1, 2 in the quad, then half-row and row mirrors.
              float s1 = (y.x + y.y) + (y.z + y.w);
              s1 += dpp_f<0xB1>(s1); s1 += dpp_f<0x4E>(s1); s1 += dpp_f<0x141>(s1); s1 += dpp_f<0x140>(s1);
              const float mp = s1 * (1.0f / 64.0f);
              const float d0 = y.x - mp, d1 = y.y - mp, d2 = y.z - mp, d3 = y.w - mp;
              float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
              q += dpp_f<0xB1>(q); q += dpp_f<0x4E>(q); q += dpp_f<0x141>(q); q += dpp_f<0x140>(q);
              if (l15 == 0) {
                if constexpr (H == 1) stat_buf[i * 16 + r] = make_float2(s1, q);
                else *reinterpret_cast<float2 *>(p.epi.fold_part + 2 * ((size_t)(ncolh >> 6) * p.M + m)) = make_float2(s1, q);
              }
            }
          } else {
            store16<nt_out>(o, v);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int r = (lane >> 3) + 8 * k, cc = (lane & 7) * 8;
          float4 v0 = *reinterpret_cast<const float4 *>(tr + r * TS + cc);
          float4 v1 = *reinterpret_cast<const float4 *>(tr + r * TS + cc + 4);
          v0.x = fin(v0.x, cadd[0][0]); v0.y = fin(v0.y, cadd[0][1]); v0.z = fin(v0.z, cadd[0][2]); v0.w = fin(v0.w, cadd[0][3]);
          v1.x = fin(v1.x, cadd[0][4]); v1.y = fin(v1.y, cadd[0][5]); v1.z = fin(v1.z, cadd[0][6]); v1.w = fin(v1.w, cadd[0][7]);
          uint4 pk;
          pk.x = (unsigned)cvt16<OutT>(v0.x) | ((unsigned)cvt16<OutT>(v0.y) << 16);
          pk.y = (unsigned)cvt16<OutT>(v0.z) | ((unsigned)cvt16<OutT>(v0.w) << 16);
          pk.z = (unsigned)cvt16<OutT>(v1.x) | ((unsigned)cvt16<OutT>(v1.y) << 16);
          pk.w = (unsigned)cvt16<OutT>(v1.z) | ((unsigned)cvt16<OutT>(v1.w) << 16);
          unsigned short *o = reinterpret_cast<unsigned short *>(p.out) + (size_t)(mrow0 + r) * ldo + ncol0 + cc;
          *reinterpret_cast<uint4 *>(o) = pk;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (SPLIT && H == 1) {
      static_assert((16 * TS + TM * 16 + 2 * TM * 16) * 4 <= SCRATCH_PER_WAVE, "folded LayerNorm: partial statistics in the wave's scratch");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (2 * lane < TM * 16) {          // two rows per lane: the wave's TM * 16 rows of plane ncol0 / 64 are contiguous
        const float4 v = *reinterpret_cast<const float4 *>(stat_buf + 2 * lane);
        *reinterpret_cast<float4 *>(p.epi.fold_part + 2 * ((size_t)(ncol0 >> 6) * p.M + m0 + wr * TM * 16 + 2 * lane)) = v;
      }
    }
  } else {
    // Generic path: edge tiles (M or N not a multiple of the tile, odd leading dimension).
    // Fully unrolled: a runtime index into acc[][] would send the accumulators to scratch.
    rejoin();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wr * TM + i) * 16 + (lane & 15);
      bool live = m < p.M;
      if ((MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) && p.epi.skip_mod > 0 && live && (m % p.epi.skip_mod) == 0) live = false;
      if (live) {
        size_t orow = (size_t)m;
        const float *posrow = nullptr, *temprow = nullptr;
        float rn = 0.f;
        if (MODE == EPI_PATCH) {
          const int np = m % p.epi.P, ft = m / p.epi.P;
          const int tt = ft % p.epi.F, item = ft / p.epi.F;
          orow = p.epi.frames_major ? (size_t)item * p.epi.T + 1 + (size_t)tt * p.epi.P + np
                                    : (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
          posrow = p.epi.pos + (size_t)(1 + np) * p.N;
          if (p.epi.temporal) temprow = p.epi.temporal + (size_t)tt * p.N;
        }
        if (MODE == EPI_L2DIST) rn = p.epi.rown[m];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wc * TN + j) * 16 + g * 4;
          OutT *o = reinterpret_cast<OutT *>(p.out) + orow * ldo + n;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e < p.N) {
              float x = acc[i][j][e];
              if (p.bias) x += p.bias[n + e];
              if (MODE == VTC_EPI_GELU) x = quick_gelu<sizeof(T) == 4>(x);
              if (MODE == EPI_PATCH) x += posrow[n + e] + (temprow ? temprow[n + e] : 0.f);
              if (MODE == EPI_L2DIST) x = (rn - 2.0f * x) + p.epi.coln[n + e];   // same rounding sequence as the interior path
              if (MODE == EPI_SCALE) x *= scale;
              if (MODE == VTC_EPI_RESID || MODE == EPI_RESID_LN) reinterpret_cast<float *>(o)[e] += x;
              else ElemOps<OutT>::store(o + e, x);
            }
          }
        }
      }
    }
  }

}

// WM x WN waves, each owning TM x TN MFMA tiles of 16x16.
// NSTAGE LDS buffers: 2 = wait for the next slab at the end of every K-step; 3 = the LDS-DMA of slab
// t+2 is issued at step t and only slab t+1 is waited for (counted s_waitcnt vmcnt(G)), so two slabs are
// always in flight and the DMA latency has two K-steps of matrix work to hide under.
template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(WM *WN * 64, 2) void gemm_kernel(GemmParams p) {
  constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * TN * 16;
  if (p.epi.m_dev) {        // the row count lives in device memory (GemmEpi::m_dev): the grid was sized for the host's upper bound
    p.M = *p.epi.m_dev;
    p.MT = (p.M + BM - 1) / BM;
  }
  constexpr int G = BM / 8 / NW + BN / 8 / NW;          // LDS-DMA instructions per wave per slab
  constexpr bool STAGGER = NW == 8;                     // two waves per SIMD inside one workgroup
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int AG = BM / 8 / NW, WG = BN / 8 / NW;
  constexpr int SUPER = SUPER_ROWS / BM;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int g = lane >> 4;

  // ---- persistent, XCD-aware tile walk ----------------------------------------------------
  // (split-K, GemmEpi::ksplit: the walk runs over (tile, K slice) pairs, slice-minor; every workgroup has exactly one)
  const int S = p.epi.ksplit > 1 ? p.epi.ksplit : 1;
  const int ntiles = p.MT * p.NT * S, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);                 // workgroups sharing this XCD label
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);           // tiles given to this XCD label
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  auto decode = [&](int logical, int &m0, int &n0) {
    if (S > 1) logical /= S;
    const int per_super = SUPER * p.NT;
    const int sr = logical / per_super, rem = logical - sr * per_super;
    const int gsz = min(SUPER, p.MT - sr * SUPER);
    const int nt = rem / gsz;
    m0 = (sr * SUPER + (rem - nt * gsz)) * BM;
    n0 = nt * BN;
  };
  int li = slot;
  if (li >= nt_x) return;                      // uniform for the whole workgroup
  int m0, n0;
  decode(start_x + li, m0, n0);
  if (S > 1) {                                   // this workgroup's K slice: shifted operand bases, its own partial-product plane
    const int ks = (start_x + li) % S;
    p.K /= S;
    p.A += (size_t)ks * p.K * sizeof(T);
    p.W += (size_t)ks * p.K * sizeof(T);
    p.out = reinterpret_cast<float *>(p.out) + (size_t)ks * p.epi.split_stride;
  }

  const int ksteps = p.K / Mma<T>::KPR;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int swz = (lane >> 1) & 7;
  const int arow = (wr * TM * 16 + (lane & 15)) * ROWB;
  const int wrow = (wc * TN * 16 + (lane & 15)) * ROWB;
  const int ldo = p.ldo;
  const bool vec_ok = (ldo & 3) == 0;

  int cur = 0;                                   // LDS buffer of the slab being multiplied
  // both operands addressable with 32-bit byte offsets (true for every tower shape)?
  const bool small32 = (size_t)p.M * p.lda_bytes < (1ull << 32) && (size_t)p.N * p.ldw_bytes < (1ull << 32);
  unsigned a_off[AG], w_off[WG];
  tile_offsets<AG>(a_off, m0, p.M, p.lda_bytes, wave_u, lane);
  tile_offsets<WG>(w_off, n0, p.N, p.ldw_bytes, wave_u, lane);
  int m0n = 0, n0n = 0;
  bool has_next = li + nb_x < nt_x;
  if (has_next) decode(start_x + li + nb_x, m0n, n0n);
  // stage the slab that is `d` K-steps after step t of the current tile (it may belong to the next tile)
  auto stage_ahead = [&](int t, int d) -> bool {
    const unsigned dst = lds_base + ((cur + d) % NSTAGE) * STAGE;
    const int k = t + d;
    if (k < ksteps) {
      if (small32) {                             // uniform base + precomputed 32-bit lane offsets
        stage_tile_fast<AG>(a_off, p.A + (size_t)k * ROWB, dst, wave_u);
        stage_tile_fast<WG>(w_off, p.W + (size_t)k * ROWB, dst + A_BYTES, wave_u);
      } else {
        stage_tile<AG>(p.A, m0, p.M, p.lda_bytes, k * ROWB, dst, wave_u, lane);
        stage_tile<WG>(p.W, n0, p.N, p.ldw_bytes, k * ROWB, dst + A_BYTES, wave_u, lane);
      }
      return true;
    }
    if (has_next) {                              // rides under the last K-steps of this tile
      stage_tile<AG>(p.A, m0n, p.M, p.lda_bytes, (k - ksteps) * ROWB, dst, wave_u, lane);
      stage_tile<WG>(p.W, n0n, p.N, p.ldw_bytes, (k - ksteps) * ROWB, dst + A_BYTES, wave_u, lane);
      return true;
    }
    return false;
  };
  // `ahead` = slabs in flight or landed BEHIND the one being multiplied (<= NSTAGE - 1).  A wait that must be sure of the next slab may
  // leave the ahead - 1 slabs issued after it in flight (vmcnt is in order): NSTAGE - 2 slabs of LDS-DMA cover the L2 / HBM round trip
  // of these launches, whose time is one tile's serial K loop (2 - 5 stages; the 64 x 64 configuration runs 3).
  auto wait_all_but = [&](int slabs) {       // uniform; slabs <= NSTAGE - 2
    if constexpr (NSTAGE >= 3) {
      if (slabs >= 1) {
        if constexpr (NSTAGE >= 4) {
          if (slabs >= 2) {
            if constexpr (NSTAGE >= 5) {
              if (slabs >= 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * G) : "memory"); return; }
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
            return;
          }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        return;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(NSTAGE >= 2 && NSTAGE <= 5 && 3 * G <= 63, "stages / counted waits");
  int ahead = -1;
#pragma unroll
  for (int d = 0; d < NSTAGE - 1; ++d)
    if (stage_ahead(0, d)) ++ahead;
  wait_all_but(ahead);                       // slab 0 has landed
  __syncthreads();

  [[maybe_unused]] float hr[TM], hc[TN][4];       // EPI_L2MIN2: -(norms) / 2 of the tile about to start (l2min2_half_norms)
  if constexpr (l2min_half_keys(MODE) && TN == 4) l2min2_half_norms<WM, WN, TM, TN>(p, m0, n0, hr, hc);
  while (true) {

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (l2min_half_keys(MODE) && TN == 4) acc[i][j] = (f32x4){hr[i] + hc[j][0], hr[i] + hc[j][1], hr[i] + hc[j][2], hr[i] + hc[j][3]};
        else acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }

    for (int t = 0; t < ksteps; ++t) {
      // Stagger (MI355X_MICROARCH "two waves per SIMD", item 9): the two waves that share a SIMD run the
      // same program between the same barriers, so left alone they issue their LDS-DMA pieces (~100
      // cycles of issue each, no matrix work) at the same time and their MFMAs at the same time.  The
      // second half of the workgroup issues its DMA after the first half-step instead, so one wave's
      // address/DMA issue sits beside the other's MFMAs.
      const bool late_dma = STAGGER && wave_u >= NW / 2;
      bool issued = false;
      if (!late_dma) issued = stage_ahead(t, NSTAGE - 1);
      // Fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt waits: hipcc, left to itself,
      // keeps ONE A-fragment register in this loop (read -> lgkmcnt(0) -> 4 MFMA -> read ...), which parks
      // every wave on an LDS round trip per 4 MFMAs (SQ_WAIT_ANY 42 %, MFMA busy 44 %).  Here the read of
      // fragment i+1 is in flight while the MFMAs of fragment i issue (LDS returns in order, so
      // lgkmcnt(1) == "everything but the youngest read has landed").
      const unsigned a_addr = lds_base + cur * STAGE + arow;
      const unsigned w_addr = lds_base + cur * STAGE + A_BYTES + wrow;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && late_dma) issued = stage_ahead(t, NSTAGE - 1);
        const unsigned coff = ((4 * ks + g) ^ swz) << 4;
        u32x4 wf[TN], aE, aO;
#pragma unroll
        for (int j = 0; j < TN; ++j) lds_read16(wf[j], w_addr + coff, j * 16 * ROWB);
        lds_read16(aE, a_addr + coff, 0);
#pragma unroll
        for (int i = 0; i < TM; i += 2) {
          lds_read16(aO, a_addr + coff, (i + 1) * 16 * ROWB);
          if (i == 0) lgkm_wait_frags<1>(aE, wf);
          else lgkm_wait<1>(aE);
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aE, acc[i][j]);
          if (i + 2 < TM) {
            lds_read16(aE, a_addr + coff, (i + 2) * 16 * ROWB);
            lgkm_wait<1>(aO);
          } else {
            lgkm_wait<0>(aO);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aO, acc[i + 1][j]);
        }
      }
      if (issued) ++ahead;
      wait_all_but(ahead - 1);               // the next slab has landed; the slabs issued after it may stay in flight
      --ahead;
      __syncthreads();
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }

    if constexpr (l2min_half_keys(MODE) && TN == 4) {
      if (has_next) l2min2_half_norms<WM, WN, TM, TN>(p, m0n, n0n, hr, hc);
    }
    tile_epilogue<T, MODE, OutT, WM, WN, TM, TN, STAGE / NW>(acc, p, m0, n0, ((cur + NSTAGE - 1) % NSTAGE) * STAGE);
    if (has_next) __syncthreads();         // the transposition area becomes the next K-step's staging buffer

    if (!has_next) break;
    li += nb_x; m0 = m0n; n0 = n0n;
    tile_offsets<AG>(a_off, m0, p.M, p.lda_bytes, wave_u, lane);
    tile_offsets<WG>(w_off, n0, p.N, p.ldw_bytes, wave_u, lane);
    has_next = li + nb_x < nt_x;
    if (has_next) decode(start_x + li + nb_x, m0n, n0n);
  }
}


// ---- EPI_RESID_LN tail: the LayerNorm that follows a residual GEMM, done by the LAST column tile of a 256-row block ---------
// Hand-off without fences (cdna_hip_programming.md Guideline 16 R1; MI355X_MICROARCH "Valid forms", first table row): every
// tile stores its part of the residual stream WRITE-THROUGH (sc1); each storing wave drains (vmcnt(0)), the workgroup meets at
// a barrier, one lane adds 1 to the row block's agent-scope counter; the workgroup whose add returns NT - 1 knows that every
// column tile of the block is stored and drained, and reads the rows back with sc1 loads ONLY (they bypass this CU's L1, the one
// cache that can hold stale copies; no line of x is shared between column tiles: 256 columns = 8 whole lines per row).
// Nothing waits on anything: the other tiles just leave.  Rows come 8 at a time per wave (up to 32 16-byte loads in flight).
// (noinline: its registers are allocated apart from the K loop's, which has none to spare)
template <typename T16>
__device__ __attribute__((noinline)) void fused_ln_rows(const float *out, int M, int N, int m0, const float *ln_g, const float *ln_b,
                                                        T16 *ln_out) {
  typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(out), 0, (int)((size_t)M * N * 4), 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int width = N;
  const int rows = min(256, M - m0);
  constexpr int RB = 8;
  for (int r0 = wave; r0 < rows; r0 += 8 * RB) {
    // all loads of the batch are issued before the first is used: no control flow around them (rows / chunks that do not exist
    // re-read an existing one and are dropped at the compute step)
    float4 v[RB][LN_MAXV][2];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int r = min(r0 + 8 * b, rows - 1);
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        const int c = min((lane + 64 * i) * 8, width - 8);
        const int off = (int)(((size_t)(m0 + r) * width + c) * 4);
        const v4u_t a = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 16);        // aux 16 = sc1
        const v4u_t b4 = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off + 16, 0, 16);
        v[b][i][0] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        v[b][i][1] = make_float4(__uint_as_float(b4.x), __uint_as_float(b4.y), __uint_as_float(b4.z), __uint_as_float(b4.w));
      }
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      const int r = r0 + 8 * b;
      if (r < rows) ln_row_compute<T16, false>(v[b], ln_g, ln_b, ln_out + (size_t)(m0 + r) * width, width, lane);
    }
  }
}

// =====================================================================================================
// Phased 256x256 bf16 kernel -- the tower shapes (every problem that fills the chip with 256x256 tiles).
//
// Same tile, LDS image, swizzle, persistent XCD-aware walk and epilogue as gemm_kernel; what differs is the
// K loop.  gemm_kernel's waves run free between one barrier per K-step: each issues its LDS reads, its MFMAs and
// its share of the next slab's LDS-DMA whenever its own program gets there.  Measured on that loop (ablation
// builds, 8192^3): 1120 TFLOP/s as is, 1449 without the DMA, 1242 without the LDS reads, 1852 with neither --
// the matrix pipe idles half the time because LDS reads, DMA landing (LDS writes) and MFMA issue of eight
// unsynchronised waves interfere, and moving the DMA issue around (top of step, interleaved, one loader wave per
// SIMD) changes nothing.  Here a K-tile is FOUR phases, one 64x32 output quadrant x K=64 each:
//     {issue the quadrant's new fragment reads; issue 2 LDS-DMA pieces (1/4 of the next K-tile);
//      s_barrier; lgkmcnt(0); setprio 1; 16 MFMA; setprio 0; vmcnt(2); s_barrier}
// with waves 4-7 running ONE BARRIER BEHIND waves 0-3: on every SIMD one wave is inside an uninterrupted MFMA
// cluster while its partner reads LDS and issues DMA, and they swap at every barrier.  The next K-tile streams
// in a quarter per phase with one quarter always in flight across the barriers (counted vmcnt, never 0).  Quadrant walk (0,0) (0,1) (1,1) (1,0): 12 / 4 / 8 / 4 fragment reads; a quarter is the
// 128 activation rows or weight rows the next K-tile's phase needs first: A0, W0, W1, A1.
// (structure after the 8-phase schedule of the CDNA HIP guide, section 5.)
//
// DEEP (round 4) -- the same phases on a deeper LDS-DMA pipeline, after the guide's template ("three half-tiles in flight, counted
// vmcnt once per K-tile"): a quarter is issued THREE to FIVE phases before the wait that retires it instead of one, so a quarter that
// misses the L2 (every activation quarter's first touch does) no longer parks the workgroup.  With two stages the depth comes from
// re-filling the stage that is being READ, quarter by quarter, as the walk frees it:
//     K-tile t reads stage X (t even/odd), stage Y holds K-tile t+1:
//       ph0: read A0, W0 (12)   issue W0(t+1) -> Y
//       ph1: read W1 (4)        issue A1(t+1) -> Y   vmcnt(8): A1(t) has landed            [read in ph2]
//       ph2: read A1 (8)        issue A0(t+2) -> X   (A0 of X: last read in ph0)
//       ph3: read W0 again (4), retired BEFORE the phase's first barrier
//                               issue W1(t+2) -> X   vmcnt(6): A0, W0, W1 of K-tile t+1 have landed  [read in ph0 / ph1 of t+1]
//     (W1 of X: last read in ph1).
// Rules kept (guide 5, "Read a staged buffer one phase AFTER the wait that retires it"; WAR: "restage >= 2 phases after the last
// ds_read, or 1 phase after when an lgkmcnt before the reading phase's first barrier retired those reads"): every wait sits before
// the first barrier of its phase and its data is read in a later phase (the lagging half of the workgroup has then waited too);
// every quarter is re-filled two phases after its last read (W0: one phase, behind the early lgkmcnt of ph3).
// Tile boundaries: the last K-tile of a tile issues nothing in ph2 / ph3 (its stage X is the epilogue's transposition scratch), and
// the first K-tile of the next tile issues those two quarters on top of its own in ph0 / ph1 -- the in-order vmcnt counts come out
// the same (8 and 6), plus the epilogue's NST stores in the relaxed first wait.
template <int MODE, typename OutT, typename T, int DEEP = 0>
__global__ __launch_bounds__(512, 2) void gemm_phased_kernel(GemmParams p) {
  static_assert(sizeof(T) == 2, "16-bit operands (bf16 or IEEE half)");
  static_assert(DEEP == 0 || DEEP == 1, "0 = one quarter in flight, 1 = deep (W0 re-read)");
  constexpr int WM = 2, WN = 4, TM = 8, TN = 4, NW = 8, BM = 256, BN = 256;
  if (p.epi.m_dev) {        // the row count lives in device memory (GemmEpi::m_dev): the grid was sized for the host's upper bound
    p.M = *p.epi.m_dev;
    p.MT = (p.M + BM - 1) / BM;
  }
  constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;
  const int SUPER = p.super_tiles > 0 ? p.super_tiles : SUPER_ROWS / BM;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4;

  // ---- persistent, XCD-aware tile walk (as gemm_kernel) ----
  const int ntiles = p.MT * p.NT, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  // Logical order: column groups of `cg` column tiles (all of them when col_group == 0); inside a group super-rows of SUPER
  // row blocks, inside a super-row column by column.  A group's weight panels are what an XCD keeps in its L2 while it walks
  // down the rows.
  const int cg = p.col_group > 0 ? min(p.col_group, p.NT) : p.NT;
  auto decode = [&](int logical, int &m0, int &n0) {
    const int full = p.MT * cg;
    const int gi = logical / full;
    const int c0 = gi * cg, cgi = min(cg, p.NT - c0);
    const int in_g = logical - gi * full;
    const int per_super = SUPER * cgi;
    const int sr = in_g / per_super, rem = in_g - sr * per_super;
    const int gsz = min(SUPER, p.MT - sr * SUPER);
    const int nt = rem / gsz;
    m0 = (sr * SUPER + (rem - nt * gsz)) * BM;
    n0 = (c0 + nt) * BN;
  };
  int li = slot;
  if (li >= nt_x) return;                      // uniform for the whole workgroup
  if (p.stagger_groups > 1) {
    const long long t_end = (long long)__builtin_amdgcn_s_memrealtime() + (long long)(slot % p.stagger_groups) * p.stagger_ticks;
    while ((long long)__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
  }
  int m0, n0;
  decode(start_x + li, m0, n0);
  int m0n = 0, n0n = 0;
  bool has_next = li + nb_x < nt_x;
  if (has_next) decode(start_x + li + nb_x, m0n, n0n);

  const int ksteps = p.K / 64;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int swz = (lane >> 1) & 7;
  // fragment read addresses inside a stage: activation fragment i = rows wr*128 + 16 i .., weight fragment j
  const unsigned a_rd = (wr * 128 + (lane & 15)) * ROWB;
  const unsigned w_rd = A_BYTES + (wc * 64 + (lane & 15)) * ROWB;
  const unsigned coff0 = ((0 + g) ^ swz) << 4, coff1 = ((4 + g) ^ swz) << 4;
  // The four fragment read addresses of the stage being read live in registers of their own and move to the other stage once per
  // K-tile (four vector adds): formed per use from (stage, a_rd, coff) they were ~10 vector adds per K-tile in the memory parts of the
  // phases, where every vector instruction delays the partner wave's MFMA issue (profiles/r04_experiments.txt 15).
  unsigned ra0 = lds_base + a_rd + coff0, ra1 = lds_base + a_rd + coff1, rw0 = lds_base + w_rd + coff0, rw1 = lds_base + w_rd + coff1;
  int rd_step = STAGE;     // to the other stage and back

  // ---- LDS-DMA quarters ----
  // first 8-row group of this wave's two (consecutive) pieces of quarter A0 / W0; A1 = +8 groups, W1 = +4
  const int ga0 = (wave_u >> 2) * 16 + (wave_u & 3) * 2;
  const int gw0 = (wave_u >> 1) * 8 + (wave_u & 1) * 2;
  // per-lane source offset of a piece whose group index is even; the odd one (second piece) uses ^ 64
  const unsigned a_vo = (unsigned)(lane >> 3) * (unsigned)p.lda_bytes + (((lane & 7) ^ (lane >> 4)) << 4);
  const unsigned w_vo = (unsigned)(lane >> 3) * (unsigned)p.ldw_bytes + (((lane & 7) ^ (lane >> 4)) << 4);
  // second piece of a quarter: eight rows further in the source, 1 KiB further in LDS (the instruction offset moves both)
  const unsigned a_vo2 = (a_vo ^ 64u) + 8u * (unsigned)p.lda_bytes - 1024u, w_vo2 = (w_vo ^ 64u) + 8u * (unsigned)p.ldw_bytes - 1024u;
  // byte offsets of this wave's row groups inside a tile's rows: A0, A1 / W0, W1
  const unsigned offA0 = (unsigned)ga0 * 8u * (unsigned)p.lda_bytes, offA1 = (unsigned)(ga0 + 8) * 8u * (unsigned)p.lda_bytes;
  const unsigned offW0 = (unsigned)gw0 * 8u * (unsigned)p.ldw_bytes, offW1 = (unsigned)(gw0 + 4) * 8u * (unsigned)p.ldw_bytes;
  // first row of the current and of the next tile's operands
  const char *tbA = p.A + (size_t)m0 * p.lda_bytes, *tbW = p.W + (size_t)n0 * p.ldw_bytes;
  const char *tbAn = p.A + (size_t)m0n * p.lda_bytes, *tbWn = p.W + (size_t)n0n * p.ldw_bytes;
  // Two pieces: 16 rows x 128 B of operand `base` starting at tile row (row0 + 8 grp), K byte kb, into
  // lds_dst .. +2 KiB.  Interior tiles: wave-uniform 64-bit base + 32-bit lane offset, one M0 write (the
  // instruction offset moves both the source and the LDS destination; the second base is pre-decremented).
  // (Scalar work per call matters: the K loop loses ~0.1 % per scalar instruction added to a phase's memory part -- profiles/
  //  r04_experiments.txt 15.  Hence: the tile's row base `tile_base` is worked out once per tile (one 64-bit multiply), the wave's
  //  row-group offset `grp_off` once per kernel, and the second piece shares the first piece's scalar base -- its "+ 8 rows - 1024"
  //  sits in the lane offset `vo2`.)
  auto stage2 = [&](const char *base, const char *tile_base, unsigned grp_off, int row0, int nrows, int ld_bytes, unsigned vo, unsigned vo2, int grp,
                    int kb, unsigned lds_dst, bool fast) __attribute__((always_inline)) {
    if (__builtin_expect(fast, 1)) {
      const char *sb0 = tile_base + (grp_off + (unsigned)kb);
      asm volatile(
          "s_mov_b32 m0, %3\n\t"
          "s_nop 0\n\t"
          "global_load_lds_dwordx4 %0, %2\n\t"
          "global_load_lds_dwordx4 %1, %2 offset:1024"
          :
          : "v"(vo), "v"(vo2), "s"(sb0), "s"(lds_dst)
          : "memory");
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = (grp + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;         // clamp: tail rows re-read a valid row, never stored
        glds16(base + (size_t)gr * ld_bytes + kb + c * 16, __builtin_amdgcn_readfirstlane(lds_dst + q * 1024));
      }
    }
  };
  // ---- patch gather (EPI_PATCH with epi.gather): activation rows are patches of the pixel tensor, read in place ----
  // Per lane, the byte offsets of its four (piece, row) sources of the tile being staged (quarter A0 / A1 x two pieces):
  // patch origin + the 16-byte chunk's place inside the patch's two (patch 32) or four (patch 16) pixel rows of this K-tile.
  constexpr bool CAN_GATHER = MODE == EPI_PATCH && sizeof(T) == 2;
  const bool gather = CAN_GATHER && p.epi.gather;
  [[maybe_unused]] unsigned gvo[4] = {0, 0, 0, 0};
  [[maybe_unused]] const int g_cs = p.epi.patch == 32 ? 2 : 1;                  // log2(16-byte chunks per pixel row of a patch)
  auto gather_offsets = [&](int sm) __attribute__((always_inline)) {
    if constexpr (CAN_GATHER) {
      const int P = p.epi.P, G = p.epi.grid, res = p.epi.res, ps = p.epi.patch;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int m = min(sm + (ga0 + 8 * h + q) * 8 + (lane >> 3), p.M - 1);   // tail rows re-read a valid patch, never stored
          const int f = m / P, np = m - f * P;
          const int gy = np / G, gx = np - gy * G;
          const int c = ((lane & 7) ^ (lane >> 4)) ^ (q << 2);                   // the piece's swizzled chunk (odd groups: ^ 4)
          const unsigned origin = (unsigned)(((f * 3) * res + gy * ps) * res + gx * ps) * 2u;
          gvo[2 * h + q] = origin + (unsigned)((c >> g_cs) * res * 2 + ((c & ((1 << g_cs) - 1)) << 4));
        }
    }
  };
  // uniform byte offset of K-tile kk (64 consecutive k = (c, i, j..j+63)): channel plane + first pixel row
  auto gather_kbase = [&](int kk) -> size_t {
    const int k = kk * 64, pp = p.epi.patch * p.epi.patch;
    const int c = k / pp, i = (k - c * pp) / p.epi.patch;
    return ((size_t)c * p.epi.res + i) * p.epi.res * 2;
  };
  // quarter qi (0 = A0, 1 = W0, 2 = W1, 3 = A1) of K-tile kk of the tile at (sm, sn) into stage st
  auto stage_quarter = [&](int qi, int sm, int sn, int kk, unsigned st, bool fastA, bool fastW, const char *ta, const char *tw) __attribute__((always_inline)) {
    if (qi == 0 || qi == 3) {
      const int grp = ga0 + (qi == 3 ? 8 : 0);
      if constexpr (CAN_GATHER) {
        if (gather) {
          const char *sb0 = reinterpret_cast<const char *>(p.A) + gather_kbase(kk);
          const char *sb1 = sb0 - 1024;
          const int h = qi == 3 ? 1 : 0;
          asm volatile(
              "s_mov_b32 m0, %4\n\t"
              "s_nop 0\n\t"
              "global_load_lds_dwordx4 %0, %2\n\t"
              "global_load_lds_dwordx4 %1, %3 offset:1024"
              :
              : "v"(gvo[2 * h]), "v"(gvo[2 * h + 1]), "s"(sb0), "s"(sb1), "s"(st + grp * 1024)
              : "memory");
          return;
        }
      }
      stage2(p.A, ta, qi == 3 ? offA1 : offA0, sm, p.M, p.lda_bytes, a_vo, a_vo2, grp, kk * ROWB, st + grp * 1024, fastA);
    } else {
      const int grp = gw0 + (qi == 2 ? 4 : 0);
      stage2(p.W, tw, qi == 2 ? offW1 : offW0, sn, p.N, p.ldw_bytes, w_vo, w_vo2, grp, kk * ROWB, st + A_BYTES + grp * 1024, fastW);
    }
  };

  int cur = 0;
  {   // prologue: the whole first K-tile
    const bool fa = m0 + BM <= p.M, fw = n0 + BN <= p.N;
    if (gather) gather_offsets(m0);
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) stage_quarter(qi, m0, n0, 0, lds_base, fa, fw, tbA, tbW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  u32x4 aS[4][2], wS[2][2];     // register subtile: 4 activation x 2 weight fragments x 2 K halves
  [[maybe_unused]] u32x4 wD[1][2][2];   // the deep loop's weight fragments
  bool relax_first = false;     // the previous tile's epilogue issued exactly NST stores last (see the phase-end wait)
  // EPI_L2MIN2: -(norms) / 2 of a tile's 256 rows and 256 columns travel through 2 KiB of LDS behind the two stages -- one value per thread,
  // requested before the previous tile's epilogue and written behind it (held in registers across the epilogue, 24 per lane, they cost 4 spills)
  [[maybe_unused]] float *nrm = reinterpret_cast<float *>(lds + 2 * STAGE);
  [[maybe_unused]] auto fetch_norm = [&](int tm0, int tn0) -> float {
    if (tid < 256) {
      const int m = tm0 + tid;
      return m < p.M ? -0.5f * p.epi.rown[m] : 0.f;
    }
    const int n = tn0 + tid - 256;
    return n < p.N ? -0.5f * p.epi.coln[n] : 0.f;
  };
  if constexpr (l2min_half_keys(MODE)) {
    nrm[tid] = fetch_norm(m0, n0);
    __syncthreads();
  }
  VTC_STAMP_INIT();
  while (true) {
    // PING-PONG: waves 4-7 (the SIMD partners of waves 0-3) run one barrier behind, so that on every SIMD one
    // wave is in its MFMA cluster while the other issues its fragment reads and DMA pieces.  (Re-joined before
    // the epilogue: the transposition scratch is the stage the lagging half reads last.)
    if (wr == 1) __builtin_amdgcn_s_barrier();
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (l2min_half_keys(MODE)) {     // (l2min2_half_norms' layout: row 16 i + (lane & 15) of the wave's 128, columns 16 j + 4 g .. + 3 of its 64)
          const float hr = nrm[wr * 128 + 16 * i + (lane & 15)];
          const float4 hc = *reinterpret_cast<const float4 *>(nrm + 256 + wc * 64 + 16 * j + 4 * g);
          acc[i][j] = (f32x4){hr + hc.x, hr + hc.y, hr + hc.z, hr + hc.w};
        } else {
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }

    // The K loop exists twice: ALL_FAST = every quarter it stages (this tile's and, at its end, the next tile's first K-tile) lies
    // inside the operands, so the per-quarter "interior?" test, its two branches (one of them taken) and the clamped slow path are not
    // in the instruction stream at all -- every tile of the towers; the other copy keeps them for edge tiles.
    auto kloop = [&](auto all_fast_c) __attribute__((always_inline)) {
    constexpr bool ALL_FAST = decltype(all_fast_c)::value;
    for (int t = 0; t < ksteps; ++t) {
      const unsigned st_cur = lds_base + cur * STAGE, st_nxt = lds_base + (cur ^ 1) * STAGE;
      if constexpr (DEEP == 0) {
        // what streams in during this K-tile: K-tile t+1 of this tile, or K-tile 0 of the next tile (when this is
        // the workgroup's last tile: K-tile 0 of this one again, into the stage nobody reads any more)
        const int kn = t + 1;
        const bool to_next = kn == ksteps && has_next;
        const int sm = to_next ? m0n : m0, sn = to_next ? n0n : n0, kk = kn < ksteps ? kn : 0;
        const bool fastA = ALL_FAST || sm + BM <= p.M, fastW = ALL_FAST || sn + BN <= p.N;
        const char *ta = to_next ? tbAn : tbA, *tw = to_next ? tbWn : tbW;
        if constexpr (CAN_GATHER) {
          if (gather && to_next) gather_offsets(m0n);     // from here on the next tile's K-tile 0 streams in
        }
        static_for<4>([&](auto ph_c) __attribute__((always_inline)) {
          constexpr int ph = decltype(ph_c)::value;
          constexpr int qm = ph >> 1, qn = (ph == 1 || ph == 2) ? 1 : 0;
          // (a) this quadrant's new fragments (published by the wait + barrier that ended the previous phase)
          if constexpr (ph != 2) {
  #pragma unroll
            for (int j = 0; j < 2; ++j) {
              lds_read16(wS[j][0], rw0, (qn * 2 + j) * 16 * ROWB);
              lds_read16(wS[j][1], rw1, (qn * 2 + j) * 16 * ROWB);
            }
          }
          if constexpr (ph == 0 || ph == 2) {
  #pragma unroll
            for (int i = 0; i < 4; ++i) {
              lds_read16(aS[i][0], ra0, (qm * 4 + i) * 16 * ROWB);
              lds_read16(aS[i][1], ra1, (qm * 4 + i) * 16 * ROWB);
            }
          }
          // (b) one quarter of the next K-tile
          stage_quarter(ph, sm, sn, kk, st_nxt, fastA, fastW, ta, tw);
          // (c) everybody has issued; the reads land while we wait here
          __builtin_amdgcn_s_barrier();
          lgkm_wait_subtile(aS, wS);
          // (d) the MFMA cluster
          __builtin_amdgcn_s_setprio(VTC_MFMA_PRIO);
  #pragma unroll
          for (int ks = 0; ks < 2; ++ks)
  #pragma unroll
            for (int i = 0; i < 4; ++i)
  #pragma unroll
              for (int j = 0; j < 2; ++j) Mma<T>::run(wS[j][ks], aS[i][ks], acc[qm * 4 + i][qn * 2 + j]);
          __builtin_amdgcn_s_setprio(0);
          // (e) my pieces of every quarter but the newest have landed; the barrier makes that everybody's.  A
          //     quarter is read three phases after its issue at the earliest, and the half of the workgroup that runs
          //     one barrier ahead must not read what the other half has not waited for yet: hence one phase early.
          // First K-tile after an (interior) epilogue: the wave's NST epilogue stores are older than this K-tile's
          // pieces in the in-order vmcnt queue, and a plain vmcnt(2) here would park the wave until they are all
          // acknowledged.  Nothing issued after them is needed before the end of phase 2 (quarter A0 has one phase of
          // slack in the steady-state schedule: issued in phase 0, read after the barrier that ends phase 3, which the
          // lagging half reaches with its waits up to phase 2 done), so phases 0 and 1 leave the stores -- and the
          // quarters issued since -- in flight and only make sure of everything OLDER than the stores.
          constexpr int NST = fast_epilogue_trailing_stores<OutT, TM>();   // 16-byte stores per wave in tile_epilogue's fast paths (its LAST vm ops)
          if (ph < 2 && t == 0 && relax_first) {
            if constexpr (ph == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NST) : "memory");
          } else
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        });
      } else {
        // ---- deep pipeline (see the kernel's header comment) ----
        constexpr int QA = 1, QB = 2;                                     // quarter ids: 0 = A0, 1 = W0, 2 = W1, 3 = A1
        constexpr int NST = fast_epilogue_trailing_stores<OutT, TM>();             // 16-byte stores per wave in tile_epilogue's fast paths (its LAST vm ops)
        const bool first = t == 0, last = t == ksteps - 1;                // of this tile (ksteps >= 2: never both)
        // K-tile t+1 (-> st_nxt): of this tile, or K-tile 0 of the next tile (no next tile: K-tile 0 of this one again, into a stage
        // nobody reads any more -- the counted waits stay uniform)
        const bool nx1 = t + 1 == ksteps && has_next;
        const int sm1 = nx1 ? m0n : m0, sn1 = nx1 ? n0n : n0, kk1 = t + 1 < ksteps ? t + 1 : 0;
        const bool fA1 = ALL_FAST || sm1 + BM <= p.M, fW1 = ALL_FAST || sn1 + BN <= p.N;
        const char *ta1 = nx1 ? tbAn : tbA, *tw1 = nx1 ? tbWn : tbW;
        // K-tile t+2 (-> st_cur, behind this K-tile's reads); t = ksteps - 2: K-tile 0 of the next tile; t = ksteps - 1: nothing
        const bool nx2 = t + 2 >= ksteps && has_next;
        const int sm2 = nx2 ? m0n : m0, sn2 = nx2 ? n0n : n0, kk2 = t + 2 < ksteps ? t + 2 : 0;
        const bool fA2 = ALL_FAST || sm2 + BM <= p.M, fW2 = ALL_FAST || sn2 + BN <= p.N;
        const char *ta2 = nx2 ? tbAn : tbA, *tw2 = nx2 ? tbWn : tbW;
        static_for<4>([&](auto ph_c) __attribute__((always_inline)) {
          constexpr int ph = decltype(ph_c)::value;
          constexpr int qm = ph >> 1, qn = (ph == 1 || ph == 2) ? 1 : 0;
          constexpr int wreg = 0;
          // (a) this quadrant's new fragments
          if constexpr (ph == 0 || ph == 1 || ph == 3) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              lds_read16(wD[wreg][j][0], rw0, (qn * 2 + j) * 16 * ROWB);
              lds_read16(wD[wreg][j][1], rw1, (qn * 2 + j) * 16 * ROWB);
            }
          }
          if constexpr (ph == 0 || ph == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              lds_read16(aS[i][0], ra0, (qm * 4 + i) * 16 * ROWB);
              lds_read16(aS[i][1], ra1, (qm * 4 + i) * 16 * ROWB);
            }
          }
          // (b) this phase's quarter(s); (c) the counted waits, BEFORE the phase's first barrier (their data is read in a later phase)
          if constexpr (ph == 0) {
            stage_quarter(QA, sm1, sn1, kk1, st_nxt, fA1, fW1, ta1, tw1);
            if (first) stage_quarter(0, sm1, sn1, kk1, st_nxt, fA1, fW1, ta1, tw1);        // A0 of K-tile 1: the previous tile's ph2 did not issue it
          } else if constexpr (ph == 1) {
            if (first) stage_quarter(QB, sm1, sn1, kk1, st_nxt, fA1, fW1, ta1, tw1);       // ... nor QB in its ph3 (QB before A1: ph3's count)
            stage_quarter(3, sm1, sn1, kk1, st_nxt, fA1, fW1, ta1, tw1);
            // A1 of THIS K-tile (issued four quarters ago) has landed.  First K-tile after an interior epilogue: its NST stores sit
            // between A1 and this K-tile's quarters in the in-order queue, and nothing younger than them is needed yet.
            if (first && relax_first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          } else if constexpr (ph == 2) {
            if (!last) {
              if constexpr (CAN_GATHER) {
                if (gather && nx2 && t + 2 == ksteps) gather_offsets(m0n);       // every activation quarter from here on is the next tile's
              }
              stage_quarter(0, sm2, sn2, kk2, st_cur, fA2, fW2, ta2, tw2);
            }
          } else {
            if (!last) stage_quarter(QB, sm2, sn2, kk2, st_cur, fA2, fW2, ta2, tw2);
            lgkm_wait_w4(wD[0]);                                                  // W0's second read retired before the barrier: re-filled next phase
            // A0, QA, QB of K-tile t+1 have landed (A1 and this K-tile's two quarters stay in flight)
            if (last) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          lgkm_wait_subtile(aS, wD[0]);
          VTC_PHASE_STAMP(1);
          // (d) the MFMA cluster
          __builtin_amdgcn_s_setprio(VTC_MFMA_PRIO);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) Mma<T>::run(wD[wreg][j][ks], aS[i][ks], acc[qm * 4 + i][qn * 2 + j]);
          __builtin_amdgcn_s_setprio(0);
          VTC_PHASE_STAMP(0);
          __builtin_amdgcn_s_barrier();
        });
      }
      cur ^= 1;
      ra0 += rd_step; ra1 += rd_step; rw0 += rd_step; rw1 += rd_step;
      rd_step = -rd_step;
    }
    };
    if constexpr (MODE != EPI_L2MIN && !l2min_half_keys(MODE) && MODE != EPI_PATCH && MODE != EPI_RESID_LN) {   // (those keep one copy: register budgets)
      const bool this_in = m0 + BM <= p.M && n0 + BN <= p.N, next_in = !has_next || (m0n + BM <= p.M && n0n + BN <= p.N);
      if (this_in && next_in) kloop(std::true_type{});
      else kloop(std::false_type{});
    } else {
      kloop(std::false_type{});
    }
    VTC_STAMP(0);       // K loop (incl. the stagger barrier)
    VTC_STAMP(1);       // (the re-join wait is inside the epilogue now: waves 0-3 wait for waves 4-7's last MFMA cluster AFTER requesting the
                        //  tile's bias / statistics / first residual rows)

    [[maybe_unused]] float nrm_next = 0.f;
    if constexpr (l2min_half_keys(MODE)) {
      if (has_next) nrm_next = fetch_norm(m0n, n0n);      // requested here, written to LDS behind the epilogue
    }
    tile_epilogue<T, MODE, OutT, WM, WN, TM, TN, STAGE / NW, true>(acc, p, m0, n0, (cur ^ 1) * STAGE);
    if constexpr (MODE == EPI_RESID_LN) {
      int *ticket = reinterpret_cast<int *>(lds + 2 * STAGE);       // one word behind the two stages (run_phased asks for it)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // EVERY storing wave drains its write-through stores ...
      __syncthreads();                                             // ... before the one lane that signals for all of them
      if (tid == 0) *ticket = __hip_atomic_fetch_add(p.epi.ln_cnt + m0 / BM, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();                                             // the add has returned before any wave loads
      if (*ticket == p.NT - 1)                                     // uniform: this tile completed the row block
        fused_ln_rows<T>(reinterpret_cast<const float *>(p.out), p.M, p.N, m0, p.epi.ln_g, p.epi.ln_b, reinterpret_cast<T *>(p.epi.ln_out));
    }
    VTC_STAMP_TILE_END(!has_next);   // epilogue issue
    if (!has_next) break;
    relax_first = MODE != EPI_L2MIN && !l2min_half_keys(MODE) && MODE != EPI_RESID_LN && (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((p.ldo & 3) == 0);   // the tile just stored took a fast path
    if constexpr (l2min_half_keys(MODE)) {    // (every wave read this tile's norms before its K loop: long before any wave gets here)
      nrm[tid] = nrm_next;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // the transposition area is the next K-tile's DMA target
    VTC_STAMP(3);       // post-epilogue barrier
    li += nb_x; m0 = m0n; n0 = n0n;
    tbA = tbAn; tbW = tbWn;
    has_next = li + nb_x < nt_x;
    if (has_next) {
      decode(start_x + li + nb_x, m0n, n0n);
      tbAn = p.A + (size_t)m0n * p.lda_bytes; tbWn = p.W + (size_t)n0n * p.ldw_bytes;
    }
  }
}


}  // namespace
namespace vtcgemm {
// CU count of the CURRENT device (cached per device: grids are sized for the card the launch goes to)
int num_cus() {
  static std::atomic<int> cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev].store(n = v, std::memory_order_relaxed);
  }
  return n;
}
}  // namespace vtcgemm
namespace {

template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN, int NSTAGE>
int run(GemmParams p, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT_ = WM * WN * 64;
  p.MT = cdiv(p.M, BM); p.NT = cdiv(p.N, BN);
  const int S = p.epi.ksplit > 1 ? p.epi.ksplit : 1;
  const int ntiles = p.MT * p.NT * S;
  const size_t shmem = (size_t)NSTAGE * (BM + BN) * ROWB;
  if (p.K / S / Mma<T>::KPR < NSTAGE - 1) { vtc_set_error("gemm: K=%d (/ %d slices) is too short for %d stages", p.K, S, NSTAGE); return 1; }
  const int wg_per_cu = shmem > 80 * 1024 ? 1 : 2;
  const int grid = min(ntiles, num_cus() * wg_per_cu);
  if (S > 1 && (grid < ntiles || p.K % (S * Mma<T>::KPR) != 0)) {
    vtc_set_error("gemm: split-K x %d needs one workgroup per (tile, slice) (%d > %d resident) and K %% (slices x %d) == 0 (K=%d)", S, ntiles, grid, Mma<T>::KPR, p.K);
    return 1;
  }
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&gemm_kernel<T, MODE, OutT, WM, WN, TM, TN, NSTAGE>), (int)shmem, "gemm")) return 1;
  hipLaunchKernelGGL((gemm_kernel<T, MODE, OutT, WM, WN, TM, TN, NSTAGE>), dim3(grid), dim3(NT_), shmem, stream, p);
  VTC_LAUNCH_CHECK("gemm");
  return 0;
}

int g_cu_budget = 0;                  // > 0: the persistent 256 x 256 grid takes at most this many workgroups (= CUs) -- two streams' GEMMs side by side (diagnostics: VTC_GEMM_CU_BUDGET)
int g_deep = VTC_GEMM_DEEP_DEFAULT;   // pipeline depth of the 256 x 256 kernel: 0 = one quarter in flight (rounds 1-3), 1 = deep (VTC_GEMM_DEEP)

template <int MODE, typename OutT, typename T, int DEEP>
int run_phased_d(GemmParams p, hipStream_t stream) {
  p.MT = cdiv(p.M, 256); p.NT = cdiv(p.N, 256);
  const int ntiles = p.MT * p.NT;
  const size_t shmem = (size_t)2 * 512 * ROWB + (MODE == EPI_RESID_LN ? 16 : 0) + (l2min_half_keys(MODE) ? 2048 : 0);   // 128 KiB: one workgroup per CU (+ the ticket word / the tile's half norms)
  const int grid = min(ntiles, g_cu_budget > 0 ? min(g_cu_budget, num_cus()) : num_cus());
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&gemm_phased_kernel<MODE, OutT, T, DEEP>), (int)shmem, "gemm_phased")) return 1;
  VTC_STAMP_HOST_BEFORE(p, stream);
  hipLaunchKernelGGL((gemm_phased_kernel<MODE, OutT, T, DEEP>), dim3(grid), dim3(512), shmem, stream, p);
  VTC_STAMP_HOST_AFTER(p, stream, grid, MODE);
  VTC_LAUNCH_CHECK("gemm_phased");
  return 0;
}
template <int MODE, typename OutT, typename T>
int run_phased(const GemmParams &p, hipStream_t stream) {
  // the deep pipeline needs two K-tiles per tile; the fused-LayerNorm tail and the sweep's block-minima epilogue stay on the
  // round-3 loop (register budgets: EPI_L2MIN with the deep loop spills 18 registers and measures the same, r04_experiments.txt 7)
  // (round 5: with the epilogue's thread index opaque the two-plane form fits the deep loop without spills -- 2.94 - 2.96 against 2.89 - 2.93 ms at 50k: no gain)
  if constexpr (MODE != EPI_RESID_LN && MODE != EPI_L2MIN && !l2min_half_keys(MODE)) {
    if (p.K >= 128) {
      if (g_deep >= 1) return run_phased_d<MODE, OutT, T, 1>(p, stream);
    }
  }
  return run_phased_d<MODE, OutT, T, 0>(p, stream);
}

int g_resid_small_k = 0; // residual epilogues with K <= this take the 128 x 128 kernel (two workgroups per CU: one's epilogue under the other's K loop); 0 = heuristic only (VTC_GEMM_RESID_SMALL_K)
int g_force_tile = 0;   // 0 = heuristic, 1 = 128x128, 2 = 256x256 free-running, 4 = 256x256 phased, 5 = 64x64 (diagnostics: VTC_GEMM_TILE)

template <typename T, int MODE, typename OutT>
int run_cfg(const GemmParams &p, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    // Tile choice by estimated rounds: a round of 256x256 tiles (one per CU) costs ~1.0, a round of 128x128 tiles
    // (two per CU) ~0.65 of that (calibrated on the vision-tower shapes, tools/gemm_tile_choice.py: 150 big tiles
    // -> big, 297 -> small, 75 -> small, 256 -> big, 450 and more -> big).
    const long tb = (long)cdiv(p.M, 256) * cdiv(p.N, 256), ts = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
    const long rb = (tb + num_cus() - 1) / num_cus(), rs = (ts + 2 * num_cus() - 1) / (2 * num_cus());
    bool big = rb * 100 <= rs * 65;
    if constexpr (MODE == VTC_EPI_RESID || MODE == EPI_RESID_FOLD || MODE == EPI_RESID_FOLD_C) {
      if (g_resid_small_k > 0 && p.K <= g_resid_small_k) big = false;
    }
    if (g_force_tile == 1) big = false;
    if (g_force_tile == 2) big = true;
    if (g_force_tile == 4) big = true;
    if (big && g_force_tile != 2) return run_phased<MODE, OutT, T>(p, stream);
    if (big) return run<T, MODE, OutT, 2, 4, 8, 4, 2>(p, stream);
  }
  // few 128x128 tiles (CAM: 1536 x 512, the output projections): 64x64 tiles, two waves, put 4x the workgroups
  // on the chip -- these launches are bounded by one tile's serial K loop, not by throughput
  const long ts128 = (long)cdiv(p.M, 128) * cdiv(p.N, 128);
#ifndef VTC_SMALL_NSTAGE
#define VTC_SMALL_NSTAGE 3      // LDS stages of the 64 x 64 configuration: two K-steps of LDS-DMA in flight (these launches are one tile's serial K loop: B = 1 forward 2.36 -> 1.98 ms from 2 to 3 stages in round 2; 5 stages, round 5: no further gain -- 1.83 against 1.78 - 1.80 ms, profiles/r05_experiments.txt 8)
#endif
  if ((ts128 * 2 <= num_cus() && g_force_tile == 0) || g_force_tile == 5) {
    // the slabs staged ahead may reach into the NEXT tile but not beyond it: a K loop of ksteps slabs carries at most ksteps + 1 stages
    const int ksteps = p.K / Mma<T>::KPR;
    if (ksteps >= VTC_SMALL_NSTAGE - 1) return run<T, MODE, OutT, 2, 1, 2, 4, VTC_SMALL_NSTAGE>(p, stream);
    if (ksteps >= 2) return run<T, MODE, OutT, 2, 1, 2, 4, 3>(p, stream);
    return run<T, MODE, OutT, 2, 1, 2, 4, 2>(p, stream);
  }
  return run<T, MODE, OutT, 2, 2, 4, 4, 2>(p, stream);
}

// sweep epilogue: 0 = phased 256 x 256 tiles (row blocks of 128), 1 = 128 x 128 tiles, two workgroups per CU (row blocks of 64)
int l2min_row_block(int variant) { return variant == 0 ? 128 : 64; }
int run_l2min(const GemmParams &p, hipStream_t stream) {
  // (round 5: a third form -- two 4-wave workgroups per CU on 128 x 256 tiles, K = 32 slabs, so that one's epilogue runs under the
  //  other's K loop -- was built, passes the sweep tests and measures 5.2 ms against 4.6 at 50k: tools/probes/gemm_l2min2.hip,
  //  profiles/r05_experiments.txt 3)
  if (p.epi.mode == EPI_L2MIN2) {
    if (p.epi.rb == 128) return run_phased<EPI_L2MIN2, float, bf16_t>(p, stream);
    return run<bf16_t, EPI_L2MIN2, float, 2, 2, 4, 4, 2>(p, stream);
  }
  if (p.epi.mode == EPI_L2MIN3) {
    if (p.epi.rb == 128) return run_phased<EPI_L2MIN3, float, bf16_t>(p, stream);
    return run<bf16_t, EPI_L2MIN3, float, 2, 2, 4, 4, 2>(p, stream);
  }
  if (p.epi.rb == 128) return run_phased<EPI_L2MIN, float, bf16_t>(p, stream);
  return run<bf16_t, EPI_L2MIN, float, 2, 2, 4, 4, 2>(p, stream);
}

template <typename T>
int dispatch(GemmParams p, hipStream_t stream) {
  const bool out_f32 = p.epi.out_dtype == VTC_F32;
  using Out16 = std::conditional_t<sizeof(T) == 2, T, bf16_t>;   // 16-bit outputs are in the operand format (fp32 operands: bf16)
  switch (p.epi.mode) {
    case VTC_EPI_STORE:
      if constexpr (sizeof(T) == 2) {
        if (p.epi.ksplit > 1) {      // (launch_gemm checked: fp32 partial planes, 64 x 64 tiles, one workgroup per (tile, slice))
          if (p.K / p.epi.ksplit / Mma<T>::KPR >= VTC_SMALL_NSTAGE - 1) return run<T, VTC_EPI_STORE, float, 2, 1, 2, 4, VTC_SMALL_NSTAGE>(p, stream);
          return run<T, VTC_EPI_STORE, float, 2, 1, 2, 4, 2>(p, stream);
        }
      }
      if (out_f32) return run_cfg<T, VTC_EPI_STORE, float>(p, stream);
      if constexpr (sizeof(T) == 2) {
        if (p.epi.fold_stat) return run_cfg<T, EPI_STORE_FOLD, Out16>(p, stream);
      }
      return run_cfg<T, VTC_EPI_STORE, Out16>(p, stream);
    case VTC_EPI_GELU:
      if (out_f32) return run_cfg<T, VTC_EPI_GELU, float>(p, stream);
      if constexpr (sizeof(T) == 2) {
        if (p.epi.fold_stat) return run_cfg<T, EPI_GELU_FOLD, Out16>(p, stream);
      }
      return run_cfg<T, VTC_EPI_GELU, Out16>(p, stream);
    case VTC_EPI_RESID:
      if constexpr (sizeof(T) == 2) {
        if (p.epi.y16 && p.epi.fold_stat) return run_cfg<T, EPI_RESID_FOLD_C, float>(p, stream);
        if (p.epi.y16) return run_cfg<T, EPI_RESID_FOLD, float>(p, stream);
      }
      return run_cfg<T, VTC_EPI_RESID, float>(p, stream);
    case EPI_PATCH:
      if constexpr (sizeof(T) == 2) {
        if (p.epi.gather) return run_phased<EPI_PATCH, float, T>(p, stream);     // only the phased kernel addresses patches in place
      }
      return run_cfg<T, EPI_PATCH, float>(p, stream);
    case EPI_L2DIST: return run_cfg<T, EPI_L2DIST, float>(p, stream);
    case EPI_SCALE: return run_cfg<T, EPI_SCALE, float>(p, stream);
    case EPI_L2MIN:
    case EPI_L2MIN2:
    case EPI_L2MIN3:
      if constexpr (sizeof(T) == 2 && !std::is_same<T, f16_t>::value) return run_l2min(p, stream);
      break;
    case EPI_RESID_LN:
      if constexpr (sizeof(T) == 2) {
        p.MT = cdiv(p.M, 256);
        (void)hipMemsetAsync(p.epi.ln_cnt, 0, align_up((size_t)p.MT * 4, 16), stream);   // arrival counters, zeroed every launch
        return run_phased<EPI_RESID_LN, float, T>(p, stream);
      }
      break;
  }
  vtc_set_error("gemm: unknown epilogue %d", p.epi.mode);
  return 1;
}

}  // namespace

// Patch embedding without the im2row matrix: 16-bit pixels in the operand format, conv1's kernel = stride = 16 or 32 (a
// K-tile of 64 is then whole pixel rows of one channel), 16-byte aligned pixel rows, 32-bit byte offsets.
bool gemm_patch_gather_supported(int n_frames, int grid, int patch, int res, int pixel_dtype, int dtype) {
  static const bool off = [] { const char *e = getenv("VTC_PATCH_IM2ROW"); return e && e[0] == '1'; }();
  return !off && pixel_dtype == dtype && (dtype == VTC_BF16 || dtype == VTC_F16) && (patch == 16 || patch == 32) && res == grid * patch &&
         (size_t)n_frames * 3 * res * res * 2 < (1ull << 32);
}

// The residual GEMM can take the FOLLOWING LayerNorm along (EPI_RESID_LN) when the phased 256 x 256 kernel runs it, a row
// is a whole number of column tiles and fits ln_row.h, and the output is addressable by one buffer descriptor.
bool gemm_resid_ln_supported(int M, int N, int K, int dtype) {
  if (dtype != VTC_BF16 && dtype != VTC_F16) return false;
  if (N % 256 != 0 || N > 512 * LN_MAXV || K % 64 != 0) return false;
  if ((size_t)M * N * 4 >= ((size_t)1 << 32)) return false;
  const long tb = (long)cdiv(M, 256) * cdiv(N, 256), ts = (long)cdiv(M, 128) * cdiv(N, 128);
  const long rb = (tb + num_cus() - 1) / num_cus(), rs = (ts + 2 * num_cus() - 1) / (2 * num_cus());
  return rb * 100 <= rs * 65;          // the tile heuristic of run_cfg picks the phased kernel
}

int launch_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                const GemmEpi &epi, hipStream_t stream) {
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "gemm: bad dtype %d", dtype);
  VTC_CHECK(epi.out_dtype == VTC_F32 || epi.out_dtype == (dtype == VTC_F16 ? VTC_F16 : VTC_BF16),
            "gemm: out_dtype %d does not go with operand dtype %d (16-bit outputs are in the operand format)", epi.out_dtype, dtype);
  const int esz = dtype == VTC_F32 ? 4 : 2;
  const int kpr = ROWB / esz;
  VTC_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VTC_CHECK(K % kpr == 0, "gemm: K=%d must be a multiple of %d for dtype %d", K, kpr, dtype);
  VTC_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: operands must be 16-byte aligned");
  VTC_CHECK(!(epi.out_dtype != VTC_F32 && (epi.mode == VTC_EPI_RESID || epi.mode >= EPI_PATCH)),
            "gemm: epilogue %d writes fp32 only", epi.mode);
  if (epi.mode == EPI_RESID_LN) {
    VTC_CHECK(gemm_resid_ln_supported(M, N, K, dtype), "gemm: fused residual + LayerNorm does not cover M=%d N=%d K=%d dtype=%d", M, N, K, dtype);
    VTC_CHECK(epi.ln_g && epi.ln_b && epi.ln_out && epi.ln_cnt && (epi.ldo == 0 || epi.ldo == N), "gemm: fused LayerNorm arguments");
  }
  if (epi.ksplit > 1) {
    VTC_CHECK(esz == 2 && epi.mode == VTC_EPI_STORE && epi.out_dtype == VTC_F32 && !bias && !epi.y16 && !epi.fold_stat && M % 64 == 0 && N % 64 == 0 &&
                  epi.split_stride >= M * (epi.ldo > 0 ? epi.ldo : N),
              "gemm: split-K takes 16-bit operands, a plain fp32 store without bias, M and N multiples of 64 and a plane stride >= M x ldo");
  }
  if (epi.y16 || epi.fold_stat) {     // folded LayerNorm: only the interior fast epilogues carry it
    // (the consumer may write its N columns into a wider output -- ldo > N: a column window of a projection; the producer's (hi, lo)
    // arrays are indexed with ldo too, so there ldo == N)
    VTC_CHECK(esz == 2 && M % 256 == 0 && N % 256 == 0 && (epi.ldo == 0 || epi.ldo == N || (!epi.y16 && epi.ldo > N && epi.ldo % 8 == 0)),
              "gemm: folded LayerNorm needs 16-bit operands and M, N multiples of 256 (M=%d N=%d dtype=%d)", M, N, dtype);
    VTC_CHECK(epi.y16 ? (epi.mode == VTC_EPI_RESID && epi.fold_part != nullptr && epi.y16lo != nullptr)
                      : ((epi.mode == VTC_EPI_STORE || epi.mode == VTC_EPI_GELU) && epi.out_dtype != VTC_F32 && epi.fold_s != nullptr),
              "gemm: folded LayerNorm arguments do not go with epilogue %d", epi.mode);
  }
  // diagnostics knobs, read once (C++11 static initialisation is thread-safe; never written afterwards)
  struct Env { int tile = 0, sg = 0, st = 0, cg = -1, rsk = 0, deep = VTC_GEMM_DEEP_DEFAULT, super = 0, cus = 0; };
  static const Env env = [] {
    Env v;
    if (const char *e = getenv("VTC_GEMM_TILE")) v.tile = atoi(e);
    if (const char *e = getenv("VTC_GEMM_CG")) v.cg = atoi(e);
    if (const char *e = getenv("VTC_GEMM_DEEP")) v.deep = atoi(e);
    if (const char *e = getenv("VTC_GEMM_SUPER")) v.super = atoi(e);
    if (const char *e = getenv("VTC_GEMM_RESID_SMALL_K")) v.rsk = atoi(e);
    if (const char *e = getenv("VTC_GEMM_STAGGER")) sscanf(e, "%d,%d", &v.sg, &v.st);
    if (const char *e = getenv("VTC_GEMM_CU_BUDGET")) v.cus = atoi(e);
    return v;
  }();
  g_force_tile = env.tile;
  g_resid_small_k = env.rsk;
  g_deep = env.deep;
  g_cu_budget = env.cus;
  GemmParams p;
  p.A = (const char *)A; p.W = (const char *)W; p.bias = bias; p.out = out;
  p.M = M; p.N = N; p.K = K;
  p.lda_bytes = K * esz; p.ldw_bytes = K * esz;
  p.ldo = epi.ldo > 0 ? epi.ldo : N;
  p.MT = 0; p.NT = 0;
  p.col_group = env.cg >= 0 ? env.cg : 0;
  p.super_tiles = env.super;
  p.stagger_groups = env.sg; p.stagger_ticks = env.st;
  p.epi = epi;
  ProfScope prof(dtype != VTC_F32 ? VTC_PROF_GEMM_BF16 : VTC_PROF_GEMM_F32, epi.m_dev ? 2.0 * N * K : 2.0 * M * N * K, stream,
                 epi.m_dev);   // 16-bit operand class; device row count: work per row
  prof.tag(epi.mode + (esz == 2 && (epi.mode <= VTC_EPI_RESID) && (epi.y16 || epi.fold_stat) ? (epi.y16 && epi.fold_stat ? 9 : 8) : 0), N, K);
  if (dtype == VTC_F16) return dispatch<f16_t>(p, stream);
  return dtype == VTC_BF16 ? dispatch<bf16_t>(p, stream) : dispatch<float>(p, stream);
}

extern "C" int vtc_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                        int epilogue, int out_dtype, int skip_mod, void *stream) {
  VTC_CHECK(epilogue >= VTC_EPI_STORE && epilogue <= VTC_EPI_RESID, "vtc_gemm: bad epilogue %d", epilogue);
  GemmEpi e;
  e.mode = epilogue; e.out_dtype = epilogue == VTC_EPI_RESID ? VTC_F32 : out_dtype; e.skip_mod = skip_mod;
  return launch_gemm(A, W, bias, out, M, N, K, dtype, e, (hipStream_t)stream);
}

// out(fp32 [M,N]) += A W^T + bias (rows with m % skip_mod == 0 untouched), then ln_out[M,N] (operand format) =
// LayerNorm(out) * ln_g + ln_b -- one launch; bit-identical to vtc_gemm(VTC_EPI_RESID) followed by vtc_layernorm.
extern "C" size_t vtc_gemm_resid_layernorm_workspace_bytes(int M) { return align_up((size_t)cdiv(M, 256) * 4, 256); }
extern "C" int vtc_gemm_resid_layernorm(const void *A, const void *W, const float *bias, float *out, int M, int N, int K, int dtype,
                                        int skip_mod, const float *ln_g, const float *ln_b, void *ln_out, void *ws, size_t ws_bytes,
                                        void *stream) {
  VTC_CHECK(ws && ws_bytes >= vtc_gemm_resid_layernorm_workspace_bytes(M), "gemm_resid_layernorm: workspace too small");
  GemmEpi e;
  e.mode = EPI_RESID_LN; e.out_dtype = VTC_F32; e.skip_mod = skip_mod;
  e.ln_g = ln_g; e.ln_b = ln_b; e.ln_out = ln_out; e.ln_cnt = (int *)ws;
  return launch_gemm(A, W, bias, out, M, N, K, dtype, e, (hipStream_t)stream);
}
extern "C" int vtc_gemm_resid_layernorm_supported(int M, int N, int K, int dtype) { return gemm_resid_ln_supported(M, N, K, dtype) ? 1 : 0; }
