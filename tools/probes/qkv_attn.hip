// PROBE (not part of libvtc_hip.so since round 5, ABI 6): kept with its measurements in profiles/r05_experiments.txt 1.
// Builds against vtc_amd/csrc (gemm_common.h); the towers.hip / ops.py hooks that called it are in the history (commit cbaba74 + the
// round-5 FOLD variant below).  Measured at 1 024 videos, time branch, per layer: this kernel 1.74 ms stand-alone / 1.95 ms inside the
// tower (2.09 ms with the folded LayerNorm) against 1.20 ms (folded QKV GEMM) + 0.49 ms (attention core) -- slower.
// qkv_attn.hip -- QKV projection + attention core in ONE kernel:  out = softmax(q k^T / 8 [+ causal]) v,  [q|k|v] = h W_in^T + b
//
// Replaces, for the 16-bit operand modes, the pair {QKV GEMM -> packed qkv [rows, 3W] in HBM -> attention kernel} behind
// `multi_head_attention` (model/timesformer_clip_alt.py:43-58: in_proj at :50, scaling :52, `attn` :36-40) and upstream
// nn.MultiheadAttention's in_proj + scaled-dot-product core.  The out-projection (:65) stays a GEMM: it contracts over ALL
// heads of a token, this kernel works one head at a time.  What never touches HBM any more: the packed qkv matrix
// (written + read once per attention branch: 2 x rows x 3W x 2 bytes) and the attention kernel's launch.
//
// Work decomposition: one tile = (G whole sequences, one head).
//   GEMM phase   acc[256 rows x 192 cols] = H[rows of the G sequences, W] . Wh^T, Wh = the head's 64 q-, 64 k- and 64 v-rows
//                of in_proj_weight.  Same machinery as gemm.hip: 128-byte K rows, both operands L2 -> LDS by LDS-DMA with
//                the XOR swizzle on the source address, double buffered, v_mfma_f32_16x16x32_{bf16,f16}, 8 waves (4 x 2,
//                64 x 96 per wave).  The token rows are GATHERED through the attention row map (time: the F rows of one
//                (item, patch); space: cls + the stride-F rows of one (item, frame)) by the per-lane DMA source address --
//                the reference's rearranges (timesformer_clip_alt.py:143-158) cost nothing here either.
//   QKV -> LDS   accumulators + bias, rounded to the operand format exactly as the unfused path rounds them on their way
//                through HBM (results are bit-identical to it), Q and K row-major [token][64], V transposed [d][token].
//   attention    one wave per (sequence, 16-query tile): S^T = K Q^T on the matrix cores (keys on the accumulator rows, so
//                softmax is lane-local + two xor-shuffles), P.V with the probability accumulators as the B operand -- the
//                scheme of attention.hip, with K / Q / V^T fragments coming from LDS instead of HBM.
//   output       O overwrites the unit's own Q rows in LDS; after a barrier the workgroup stores whole 128-byte
//                (token, head) lines.  cls outputs of the space branch go to cls_out in fp32 (mean over frames follows).
//
// LDS (all 160 KiB of the CU, one workgroup per CU): stage 0 [0, 56K) | stage 1 [56K, 112K); the Q/K/V^T region
// [56K, ~157K) aliases stage 1 once the K loop is done, which leaves stage 0 free: the NEXT tile's first K-slab streams in
// during the attention phase (persistent workgroups; K-steps per tile must be even so that a tile ends on stage 1).
#include "gemm_common.h"

using namespace vtcgemm;

namespace {

constexpr int BM = 256, BN = 192, NW = 8, WM = 4, WN = 2, TM = 4, TN = 6;
constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;   // 32 + 24 = 56 KiB
constexpr int AG = BM / 8 / NW, WG = BN / 8 / NW;                                    // LDS-DMA pieces per wave per slab: 4 + 3
constexpr int Q_OFF = STAGE, K_OFF = Q_OFF + BM * 128, V_OFF = K_OFF + BM * 128;
constexpr int VS = 292;                        // V^T row stride in elements: 146 dwords = 18 mod 64 -> 16 d-rows hit 16 distinct bank pairs
constexpr int B_OFF = V_OFF + 64 * VS * 2;     // the tile's 192 bias values (q | k | v slices of in_proj_bias), fp32
constexpr int T_OFF = B_OFF + BN * 4;          // per-row tables (tile-invariant): V^T slot and (sequence, token) of dense local row r
constexpr int S_OFF = T_OFF + BM * 4;          // folded LayerNorm: the tile's 192 values of s (row sums of the gamma-scaled weights), fp32
constexpr int LDS_BYTES = S_OFF + BN * 4;      // 162816 <= 163840
static_assert(LDS_BYTES <= 163840, "one workgroup owns the CU's LDS");

struct QkvAttnParams {
  const char *A;        // LayerNorm output h [rows, W], operand format
  const char *Wq;       // in_proj_weight [3W, W], operand format
  const float *bias;    // in_proj_bias [3W]  (folded LayerNorm: c = bias + W beta)
  // Folded LayerNorm (gemm.hip EPI_STORE_FOLD, towers.hip ln_proj): A = the residual stream's `hi` plane, Wq = the gamma-scaled
  // weights, and q|k|v = rstd_m (acc - mean_m s_n) + c_n with (mean, rstd) = fold_stat[2 row], s = fold_s[3W].  NULL: plain bias.
  const float *fold_stat, *fold_s;
  char *out;            // attention output [rows, W], operand format, same row map as the input
  float *cls_out;       // space branch: token 0's output goes here, [n_seq, W] fp32 (else nullptr)
  int n_seq, L, heads, causal;
  int s2, a0, a1, a2, a3, pstride;     // row map of attention.hip / vtc_attention
  int W, G, n_groups;
};

template <int N>
__device__ __forceinline__ void lgkm_wait7(u32x4 &x, u32x4 (&w)[6]) {
  asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(x), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]) : "n"(N));
}

template <typename T> __device__ __forceinline__ void mma16(const uint4 &a, const uint4 &b, f32x4 &acc);
template <> __device__ __forceinline__ void mma16<bf16_t>(const uint4 &a, const uint4 &b, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<f16_t>(const uint4 &a, const uint4 &b, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

// workgroup barrier that publishes LDS writes: the raw s_barrier builtin is no compiler fence (IntrNoMem), so it sits
// between two compiler memory barriers; __syncthreads() would also drain the vmcnt queue (the O stores must stay in flight)
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// NT = 16-key tiles per sequence (L <= 16 NT)
// U = attention units a wave keeps in flight
// FOLD = the folded-LayerNorm transform in the accumulator -> LDS pass (QkvAttnParams::fold_stat)
template <typename T, int NT, int U, bool FOLD>
__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(const QkvAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, c16 = lane & 15;
  const int L = p.L, Lp = (L + 3) & ~3, W = p.W;
  const int ld_bytes = W * 2;
  const int n_qt = (L + 15) >> 4;

  // ---- persistent, XCD-aware tile walk: tiles (group, head), heads innermost, a contiguous range per XCD label -------
  const int ntiles = p.n_groups * p.heads, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  int li = slot;
  if (li >= nt_x) return;                       // uniform for the whole workgroup

  const int ksteps = W / 64;                    // host guarantees W % 128 == 0 (even number of K-steps)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int swz = (lane >> 1) & 7;
  const unsigned a_rd = (wr * 64 + c16) * ROWB;
  const unsigned w_rd = A_BYTES + (wc * 96 + c16) * ROWB;

  // global row of token p_ of sequence s (attention.hip's affine map)
  auto row_of = [&](int s, int p_) -> int {
    const int s_hi = s / p.s2, s_lo = s - s_hi * p.s2;
    const int base = s_hi * p.a1 + s_lo * p.a2 + p.a0;
    return p_ == 0 ? base : base + 1 + s_lo * p.a3 + (p_ - 1) * p.pstride;
  };
  // per-lane LDS-DMA source offsets of a tile: activation rows gathered through the row map (rows past the tile's last
  // sequence re-read its last row: multiplied, never stored), weight rows = the head's q / k / v slices
  auto tile_offsets_of = [&](int tile, unsigned (&a_off)[AG], unsigned (&w_off)[WG]) {
    const int group = tile / p.heads, head = tile - group * p.heads;
    const int seq0 = group * p.G;
    const int nseq = min(p.G, p.n_seq - seq0);
    const int r_used = nseq * L;
#pragma unroll
    for (int q = 0; q < AG; ++q) {
      int r = (wave_u * AG + q) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      r = min(r, r_used - 1);
      const int i = r / L;
      a_off[q] = (unsigned)row_of(seq0 + i, r - i * L) * (unsigned)ld_bytes + c * 16;
    }
#pragma unroll
    for (int q = 0; q < WG; ++q) {
      const int n = (wave_u * WG + q) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((n >> 1) & 7);
      const int wrow = (n >> 6) * W + head * 64 + (n & 63);
      w_off[q] = (unsigned)wrow * (unsigned)ld_bytes + c * 16;
    }
  };
  auto stage_slab = [&](const unsigned (&a_off)[AG], const unsigned (&w_off)[WG], int k, unsigned dst) {
    stage_tile_fast<AG>(a_off, p.A + (size_t)k * ROWB, dst, wave_u);
    stage_tile_fast<WG>(w_off, p.Wq + (size_t)k * ROWB, dst + A_BYTES, wave_u);
  };

  unsigned a_off[AG], w_off[WG];
  tile_offsets_of(start_x + li, a_off, w_off);
  stage_slab(a_off, w_off, 0, lds_base);        // first tile: cold start; later tiles: issued under the attention phase

  // tile-invariant per-row tables: dense local row r = si * L + tok  ->  V^T slot si * Lp + tok, and (si, tok)
  unsigned short *tab_slot = reinterpret_cast<unsigned short *>(lds + T_OFF);
  unsigned short *tab_st = tab_slot + BM;
  if (tid < BM) {
    const int si = tid / L, tok = tid - si * L;
    tab_slot[tid] = (unsigned short)(si * Lp + tok);
    tab_st[tid] = (unsigned short)((si << 8) | tok);
  }
  auto bias_of = [&](int tile) -> float {       // the tile's q | k | v bias slices, one value per thread (192 used)
    const int head = tile % p.heads, n = min(tid, BN - 1);
    return p.bias[(n >> 6) * W + head * 64 + (n & 63)];
  };
  float bias_next = bias_of(start_x + li);
  if (tid < BN) reinterpret_cast<float *>(lds + B_OFF)[tid] = bias_next;
  auto fs_of = [&](int tile) -> float {
    const int head = tile % p.heads, n = min(tid, BN - 1);
    return FOLD ? p.fold_s[(n >> 6) * W + head * 64 + (n & 63)] : 0.f;
  };
  [[maybe_unused]] float fs_next = 0.f;
  if constexpr (FOLD) {
    fs_next = fs_of(start_x + li);
    if (tid < BN) reinterpret_cast<float *>(lds + S_OFF)[tid] = fs_next;
  }
  bool first = true;

  while (true) {
    const int tile = start_x + li;
    const int group = tile / p.heads, head = tile - group * p.heads;
    const int seq0 = group * p.G;
    const int nseq = min(p.G, p.n_seq - seq0);
    const int r_used = nseq * L;
    const bool has_next = li + nb_x < nt_x;

    // Slab 0 has landed.  The wave's vmcnt queue is in order: behind the slab's DMA pieces sit the previous tile's cls_out
    // stores and, youngest, its O stores -- exactly BM * 8 / 512 = 4 per wave (branch-free below) -- which stay in flight.
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    first = false;
    wg_barrier_lds();
    if (has_next) {                                    // consumed after the accumulator -> LDS pass (one register each across the K loop)
      bias_next = bias_of(tile + nb_x);
      if constexpr (FOLD) fs_next = fs_of(tile + nb_x);
    }
    // folded LayerNorm: (mean, rstd) of dense local row tid, requested here (the K loop's vmcnt(0) waits retire it) and held in TWO
    // registers across the K loop; behind the K loop the rows' statistics meet in stage 0, which is free until the next tile's
    // first slab is issued (holding the lane's own four rows cost 8 registers and sent 32 to scratch)
    [[maybe_unused]] float2 fst_row = make_float2(0.f, 1.f);
    if constexpr (FOLD) {
      if (tid < BM) {
        const unsigned st = tab_st[min(tid, r_used - 1)];
        fst_row = *reinterpret_cast<const float2 *>(p.fold_stat + 2 * (size_t)row_of(seq0 + (int)(st >> 8), (int)(st & 255)));
      }
    }

    // ================= GEMM phase: acc[m][n] = sum_k H[row(m)][k] Wh[n][k] =================
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ksteps; ++t) {
      const unsigned st_cur = lds_base + (t & 1) * STAGE, st_nxt = lds_base + ((t + 1) & 1) * STAGE;
      const bool late_dma = wave_u >= NW / 2;           // SIMD partners issue their DMA half a step apart (gemm.hip)
      if (!late_dma && t + 1 < ksteps) stage_slab(a_off, w_off, t + 1, st_nxt);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && late_dma && t + 1 < ksteps) stage_slab(a_off, w_off, t + 1, st_nxt);
        const unsigned coff = ((4 * ks + g) ^ swz) << 4;
        u32x4 wf[TN], aE, aO;
#pragma unroll
        for (int j = 0; j < TN; ++j) lds_read16(wf[j], st_cur + w_rd + coff, j * 16 * ROWB);
        lds_read16(aE, st_cur + a_rd + coff, 0);
#pragma unroll
        for (int i = 0; i < TM; i += 2) {
          lds_read16(aO, st_cur + a_rd + coff, (i + 1) * 16 * ROWB);
          if (i == 0) lgkm_wait7<1>(aE, wf);
          else lgkm_wait<1>(aE);
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aE, acc[i][j]);
          if (i + 2 < TM) {
            lds_read16(aE, st_cur + a_rd + coff, (i + 2) * 16 * ROWB);
            lgkm_wait<1>(aO);
          } else {
            lgkm_wait<0>(aO);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aO, acc[i + 1][j]);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // every wave is past its last read of stage 1: the Q/K/V^T region may be written

    // ================= accumulators -> LDS: Q, K row-major [token][64] (swizzled 16-byte chunks), V^T [d][slot] ==========
    {
      // V^T slots that hold no token but are read (P is exactly 0 there, so they only have to be finite): the tail behind the
      // tile's last sequence up to the key-tile overhang, and the <= 3 slots between L and Lp of every sequence
      unsigned short *vt = reinterpret_cast<unsigned short *>(lds + V_OFF);
      const int vsu = min(VS, (nseq - 1) * Lp + NT * 16);
      for (int c = (nseq - 1) * Lp + L + (tid & 7); c < vsu; c += 8) vt[(tid >> 3) * VS + c] = 0;
      if (Lp != L) {
        for (int i = tid & 7; i < nseq - 1; i += 8)
          for (int c = L; c < Lp; ++c) vt[(tid >> 3) * VS + i * Lp + c] = 0;
      }
      // bias (folded LayerNorm: c, and s) of the lane's output columns: n = wc*96 + 16j + 4g .. +3, segment n / 64 (q, k, v), d = n % 64.
      // Column block j outermost: its four (eight) column constants live in registers only while the TM row fragments pass
      // (all TN blocks' constants at once cost 24 + 24 registers next to the 96 accumulators: spills in the folded form).
      [[maybe_unused]] float2 fst[TM];
      if constexpr (FOLD) {
        float2 *stat_lds = reinterpret_cast<float2 *>(lds);          // stage 0: nobody reads it any more, nothing streams into it yet
        if (tid < BM) stat_lds[tid] = fst_row;
        wg_barrier_lds();
#pragma unroll
        for (int i = 0; i < TM; ++i) fst[i] = stat_lds[wr * 64 + 16 * i + c16];
      }
      int vslot[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) vslot[i] = tab_slot[min(wr * 64 + 16 * i + c16, BM - 1)];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = wc * 96 + 16 * j + 4 * g;      // (n >> 6) is wave-uniform per j: 0 = q, 1 = k, 2 = v
        const int seg = n >> 6, d = n & 63;
        const float4 b4 = *reinterpret_cast<const float4 *>(lds + B_OFF + n * 4);
        [[maybe_unused]] float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FOLD) s4 = *reinterpret_cast<const float4 *>(lds + S_OFF + n * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = wr * 64 + 16 * i + c16;            // dense local token index
          if (m < r_used) {
            float v0, v1, v2, v3;
            if constexpr (FOLD) {       // the two roundings per value of gemm.hip's folded epilogue: fma(rstd, fma(-mean, s, acc), c)
              const float nmu = -fst[i].x, rs = fst[i].y;
              v0 = __builtin_fmaf(rs, __builtin_fmaf(nmu, s4.x, acc[i][j][0]), b4.x);
              v1 = __builtin_fmaf(rs, __builtin_fmaf(nmu, s4.y, acc[i][j][1]), b4.y);
              v2 = __builtin_fmaf(rs, __builtin_fmaf(nmu, s4.z, acc[i][j][2]), b4.z);
              v3 = __builtin_fmaf(rs, __builtin_fmaf(nmu, s4.w, acc[i][j][3]), b4.w);
            } else {
              v0 = acc[i][j][0] + b4.x; v1 = acc[i][j][1] + b4.y; v2 = acc[i][j][2] + b4.z; v3 = acc[i][j][3] + b4.w;
            }
            const unsigned short h0 = cvt16<T>(v0), h1 = cvt16<T>(v1), h2 = cvt16<T>(v2), h3 = cvt16<T>(v3);
            if (seg < 2) {
              const unsigned rowb = m * 128, rs_ = (m >> 1) & 7;
              uint2 pk;
              pk.x = (unsigned)h0 | ((unsigned)h1 << 16);
              pk.y = (unsigned)h2 | ((unsigned)h3 << 16);
              const unsigned off = (seg ? K_OFF : Q_OFF) + rowb + ((((unsigned)d >> 3) ^ rs_) << 4) + ((d & 7) << 1);
              *reinterpret_cast<uint2 *>(lds + off) = pk;
            } else {
              vt[(d + 0) * VS + vslot[i]] = h0; vt[(d + 1) * VS + vslot[i]] = h1;
              vt[(d + 2) * VS + vslot[i]] = h2; vt[(d + 3) * VS + vslot[i]] = h3;
            }
          }
        }
      }
    }
    wg_barrier_lds();

    // the next tile's bias slice and first K-slab stream in under the attention phase (stage 0 and the bias slot are free)
    if (has_next) {
      if (tid < BN) {
        reinterpret_cast<float *>(lds + B_OFF)[tid] = bias_next;
        if constexpr (FOLD) reinterpret_cast<float *>(lds + S_OFF)[tid] = fs_next;
      }
      tile_offsets_of(tile + nb_x, a_off, w_off);
      stage_slab(a_off, w_off, 0, lds_base);
    }

    // ================= attention: one wave per (sequence, 16-query tile), TWO units in flight per wave =================
    // (a unit is a serial chain LDS read -> MFMA -> softmax -> MFMA; two independent chains interleave)
    {
      const unsigned short *vt = reinterpret_cast<const unsigned short *>(lds + V_OFF);
      // Short sequences (time attention, L = 8; L = 4): PACK = 16 / L sequences share one 16 x 16 score tile, block-diagonal
      // mask between them -- a tile of one 8-token sequence would leave three quarters of the lanes on masked entries.
      const int pack = (!p.causal && !p.cls_out && (L == 8 || L == 4)) ? 16 / L : 1;
      const int lsh = L == 8 ? 3 : 2;
      const int n_units = pack > 1 ? (nseq + pack - 1) / pack : nseq * n_qt;
      for (int u = wave_u; u < n_units; u += U * NW) {
        int si[U], qt[U], r0[U], slot0[U], qtok[U], Le[U];
        bool valid[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const int ux = u + x * NW;
          valid[x] = ux < n_units;                        // wave-uniform
          const int uc = valid[x] ? ux : u;
          if (pack > 1) { si[x] = uc * pack; qt[x] = 0; Le[x] = min(pack, nseq - si[x]) * L; }
          else { si[x] = uc / n_qt; qt[x] = uc - si[x] * n_qt; Le[x] = L; }
          r0[x] = si[x] * L; slot0[x] = si[x] * Lp; qtok[x] = qt[x] * 16 + c16;
        }
        auto frag = [&](int off, int row, int ks) -> uint4 {
          return *reinterpret_cast<const uint4 *>(lds + off + row * 128 + (((4 * ks + g) ^ ((row >> 1) & 7)) << 4));
        };
        uint4 qf[U][2];
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const int row = r0[x] + min(qtok[x], Le[x] - 1);
          qf[x][0] = frag(Q_OFF, row, 0); qf[x][1] = frag(Q_OFF, row, 1);
        }
        f32x4 sc[U][NT];
        float mx[U], sum[U], inv[U];
#pragma unroll
        for (int x = 0; x < U; ++x) mx[x] = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
          for (int x = 0; x < U; ++x) {
            sc[x][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if ((p.causal && kt > qt[x]) || kt * 16 >= Le[x]) {   // tile entirely in the future of every query, or past the sequence end
#pragma unroll
              for (int r = 0; r < 4; ++r) sc[x][kt][r] = -INFINITY;
              continue;
            }
            const int krow = r0[x] + min(kt * 16 + c16, Le[x] - 1);
            mma16<T>(frag(K_OFF, krow, 0), qf[x][0], sc[x][kt]);
            mma16<T>(frag(K_OFF, krow, 1), qf[x][1], sc[x][kt]);
          }
#pragma unroll
          for (int x = 0; x < U; ++x) {
            if ((p.causal && kt > qt[x]) || kt * 16 >= Le[x]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 16 + g * 4 + r;
              float v = sc[x][kt][r] * 0.125f;     // q * head_dim^-0.5 (timesformer_clip_alt.py:48,52); exact power of two
              if (key >= Le[x] || (p.causal && key > qtok[x]) || (pack > 1 && (key >> lsh) != (c16 >> lsh))) v = -INFINITY;
              sc[x][kt][r] = v;
              mx[x] = fmaxf(mx[x], v);
            }
          }
        }
#pragma unroll
        for (int x = 0; x < U; ++x) mx[x] = fmaxf(mx[x], __shfl_xor(mx[x], 16, 64));
#pragma unroll
        for (int x = 0; x < U; ++x) mx[x] = fmaxf(mx[x], __shfl_xor(mx[x], 32, 64));
#pragma unroll
        for (int x = 0; x < U; ++x) {
          sum[x] = 0.f;
#pragma unroll
          for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float e = __expf(sc[x][kt][r] - mx[x]);
              sc[x][kt][r] = e;
              sum[x] += e;
            }
        }
#pragma unroll
        for (int x = 0; x < U; ++x) sum[x] += __shfl_xor(sum[x], 16, 64);
#pragma unroll
        for (int x = 0; x < U; ++x) {
          sum[x] += __shfl_xor(sum[x], 32, 64);
          inv[x] = 1.0f / sum[x];
        }
        // O^T[d][query] = sum_key V^T[d][key] P[query][key]
        f32x4 o[U][4];
#pragma unroll
        for (int x = 0; x < U; ++x)
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[x][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < (NT + 1) / 2; ++kk) {
          const int k0 = 2 * kk, k1 = 2 * kk + 1;
          const int k1c = k1 < NT ? k1 : k0;
          uint4 pf[U];
#pragma unroll
          for (int x = 0; x < U; ++x) {
            pf[x].x = (unsigned)cvt16<T>(sc[x][k0][0]) | ((unsigned)cvt16<T>(sc[x][k0][1]) << 16);
            pf[x].y = (unsigned)cvt16<T>(sc[x][k0][2]) | ((unsigned)cvt16<T>(sc[x][k0][3]) << 16);
            pf[x].z = 0; pf[x].w = 0;
            if (k1 < NT) {
              pf[x].z = (unsigned)cvt16<T>(sc[x][k1c][0]) | ((unsigned)cvt16<T>(sc[x][k1c][1]) << 16);
              pf[x].w = (unsigned)cvt16<T>(sc[x][k1c][2]) | ((unsigned)cvt16<T>(sc[x][k1c][3]) << 16);
            }
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int x = 0; x < U; ++x) {
              const unsigned short *vr = vt + (dt * 16 + c16) * VS + slot0[x] + g * 4;
              const uint2 lo = *reinterpret_cast<const uint2 *>(vr + k0 * 16);
              uint2 hi = make_uint2(0, 0);
              if (k1 < NT) hi = *reinterpret_cast<const uint2 *>(vr + k1 * 16);
              mma16<T>(make_uint4(lo.x, lo.y, hi.x, hi.y), pf[x], o[x][dt]);
            }
        }
        // lane holds O[query = qtok][d = 16 dt + 4 g .. +3]
#pragma unroll
        for (int x = 0; x < U; ++x) {
          if (valid[x] && qtok[x] < Le[x]) {
            if (p.cls_out && qtok[x] == 0) {
              float *dst = p.cls_out + (size_t)(seq0 + si[x]) * W + head * 64 + g * 4;
#pragma unroll
              for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<float4 *>(dst + dt * 16) =
                    make_float4(o[x][dt][0] * inv[x], o[x][dt][1] * inv[x], o[x][dt][2] * inv[x], o[x][dt][3] * inv[x]);
            } else {
              const int row = r0[x] + qtok[x];
              const unsigned rs = (row >> 1) & 7;
#pragma unroll
              for (int dt = 0; dt < 4; ++dt) {
                const int d = 16 * dt + 4 * g;
                uint2 pk;
                pk.x = (unsigned)cvt16<T>(o[x][dt][0] * inv[x]) | ((unsigned)cvt16<T>(o[x][dt][1] * inv[x]) << 16);
                pk.y = (unsigned)cvt16<T>(o[x][dt][2] * inv[x]) | ((unsigned)cvt16<T>(o[x][dt][3] * inv[x]) << 16);
                *reinterpret_cast<uint2 *>(lds + Q_OFF + row * 128 + ((((unsigned)d >> 3) ^ rs) << 4) + ((d & 7) << 1)) = pk;
              }
            }
          }
        }
      }
    }
    wg_barrier_lds();

    // ================= O rows -> global: whole 128-byte (token, head) lines =================
    // Branch-free: every wave issues exactly BM * 8 / 512 = 4 store instructions (the next tile's first wait counts on it).
    // Lanes without a row of their own -- past the tile's last token, or a cls token whose output went to cls_out -- repeat
    // a neighbouring row's store (same bytes to the same address).
    {
#pragma unroll
      for (int c = 0; c < BM * 8 / 512; ++c) {
        const int id = tid + 512 * c;
        int row = min(id >> 3, r_used - 1);
        const int ch = id & 7;
        unsigned st = tab_st[row];
        if (p.cls_out && (st & 255) == 0) { row += 1; st += 1; }     // L >= 2 whenever cls_out is given (host check)
        const uint4 v = *reinterpret_cast<const uint4 *>(lds + Q_OFF + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        *reinterpret_cast<uint4 *>(p.out + (size_t)row_of(seq0 + (int)(st >> 8), (int)(st & 255)) * ld_bytes + head * 128 + ch * 16) = v;
      }
    }

    if (!has_next) break;
    li += nb_x;
  }
}

template <typename T, int NT, int U, bool FOLD = false>
int run(const QkvAttnParams &p, hipStream_t stream) {
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&qkv_attn_kernel<T, NT, U, FOLD>), LDS_BYTES, "qkv_attention")) return 1;
  const int tiles = p.n_groups * p.heads;
  hipLaunchKernelGGL((qkv_attn_kernel<T, NT, U, FOLD>), dim3(min(tiles, num_cus())), dim3(512), LDS_BYTES, stream, p);
  VTC_LAUNCH_CHECK("qkv_attention");
  return 0;
}

template <typename T>
int dispatch(const QkvAttnParams &p, hipStream_t stream) {
  static const int force_u = [] { const char *e = getenv("VTC_QKVA_U"); return e ? atoi(e) : 0; }();   // diagnostics
  const int nt = cdiv(p.L, 16);
  const bool two = force_u ? force_u == 2 : nt <= 2;     // short sequences: two independent chains per wave pay
  if (p.fold_stat) {        // folded LayerNorm: the TimeSformer time branch (L = frames <= 16)
    if (nt == 1) return two ? run<T, 1, 2, true>(p, stream) : run<T, 1, 1, true>(p, stream);
    vtc_set_error("qkv_attention: the folded-LayerNorm form covers sequences of at most 16 tokens (L=%d)", p.L);
    return 1;
  }
  switch (nt) {
    case 1: return two ? run<T, 1, 2>(p, stream) : run<T, 1, 1>(p, stream);
    case 2: return two ? run<T, 2, 2>(p, stream) : run<T, 2, 1>(p, stream);
    case 3: return two ? run<T, 3, 2>(p, stream) : run<T, 3, 1>(p, stream);
    case 4: return two ? run<T, 4, 2>(p, stream) : run<T, 4, 1>(p, stream);
    case 5: return two ? run<T, 5, 2>(p, stream) : run<T, 5, 1>(p, stream);
  }
  vtc_set_error("qkv_attention: sequence length %d > 80 unsupported", p.L);
  return 1;
}

}  // namespace

// true when the fused kernel covers this problem (the callers fall back to QKV GEMM + attention kernel otherwise)
bool qkv_attention_supported(int L, int heads, int W, int dtype, size_t rows) {
  return (dtype == VTC_BF16 || dtype == VTC_F16) && L >= 1 && L <= 80 && W == heads * 64 && W % 128 == 0 &&
         rows * (size_t)W * 2 < ((size_t)1 << 32) && (size_t)3 * W * W * 2 < ((size_t)1 << 32);
}

int launch_qkv_attention(const void *h, const void *w_qkv, const float *b_qkv, void *out, float *cls_out, int n_seq, int L, int heads,
                         int causal, int s2, int a0, int a1, int a2, int a3, int pstride, size_t rows, int dtype, hipStream_t stream,
                         const float *fold_stat, const float *fold_s) {
  VTC_CHECK(n_seq > 0 && L > 0 && heads > 0 && s2 > 0, "qkv_attention: bad sizes n_seq=%d L=%d heads=%d s2=%d", n_seq, L, heads, s2);
  VTC_CHECK(qkv_attention_supported(L, heads, heads * 64, dtype, rows), "qkv_attention: unsupported problem (L=%d heads=%d dtype=%d)", L, heads, dtype);
  VTC_CHECK(!cls_out || L >= 2, "qkv_attention: cls_out needs sequences of at least two tokens");
  VTC_CHECK(((uintptr_t)h & 15) == 0 && ((uintptr_t)w_qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "qkv_attention: operands must be 16-byte aligned");
  QkvAttnParams p;
  p.A = (const char *)h; p.Wq = (const char *)w_qkv; p.bias = b_qkv; p.out = (char *)out; p.cls_out = cls_out;
  VTC_CHECK((fold_stat == nullptr) == (fold_s == nullptr), "qkv_attention: folded LayerNorm needs both the row statistics and s");
  p.fold_stat = fold_stat; p.fold_s = fold_s;
  p.n_seq = n_seq; p.L = L; p.heads = heads; p.causal = causal;
  p.s2 = s2; p.a0 = a0; p.a1 = a1; p.a2 = a2; p.a3 = a3; p.pstride = pstride;
  p.W = heads * 64;
  p.G = min(32, BM / L);
  p.n_groups = cdiv(n_seq, p.G);
  const double flops = 2.0 * n_seq * L * 3.0 * p.W * p.W + 4.0 * L * L * 64 * (double)n_seq * heads;
  ProfScope prof(VTC_PROF_GEMM_BF16, flops, stream);
  return dtype == VTC_F16 ? dispatch<f16_t>(p, stream) : dispatch<bf16_t>(p, stream);
}

extern "C" int vtc_qkv_attention(const void *h, const void *w_qkv, const float *b_qkv, void *out, float *cls_out, int n_seq, int L,
                                 int heads, int causal, int s2, int a0, int a1, int a2, int a3, int pstride, long long rows, int dtype,
                                 void *stream) {
  return launch_qkv_attention(h, w_qkv, b_qkv, out, cls_out, n_seq, L, heads, causal, s2, a0, a1, a2, a3, pstride, (size_t)rows, dtype,
                              (hipStream_t)stream, nullptr, nullptr);
}
