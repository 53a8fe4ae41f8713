// sweep.hip -- the N x N retrieval sweep and the batch-level similarity / loss.
//   vtc_l2_topk     exact squared-L2 k-NN: replaces faiss.GpuIndexFlatL2.add/search in
//                   RecallAtK.compute (model/metric.py:137-146)
//   vtc_recall_hits `target in rp[:k]` counting (model/metric.py:148-160)
//   vtc_similarity  exp(logit_scale) * V @ T^T (model/model.py:369,478,504,621)
//   vtc_clip_loss   0.5 (CE(sim, arange) + CE(sim^T, arange)) (model/loss.py:18-22)
//
// Sweep pipeline (per block of query rows, the fp32 distance matrix "tiled in HBM" as
// BASELINE.json prescribes; blocks of up to 2 GiB):
//   1. row norms |q|^2, |g|^2 in fp32 (once);
//   2. distance GEMM with the L2 epilogue  d = |q|^2 + |g|^2 - 2 q.g  (gemm.hip, EPI_L2DIST):
//        F32     fp32 MFMA, exact;
//        BF16X3  operands split hi/lo into bf16 and concatenated along K
//                ([q_hi|q_lo|q_hi] . [g_hi|g_hi|g_lo]) so that ONE bf16 GEMM with K' = 3D forms
//                q_hi g_hi + q_lo g_hi + q_hi g_lo with fp32 accumulation;
//        BF16    hi parts only;
//   3. streaming top-k: one wave per query row reads the row with 16-byte loads; the wave keeps
//      the sorted best list ONE ENTRY PER LANE; a value is compared against the current k-th best
//      (wave-uniform), so after warm-up almost every step is load + compare + ballot; the rare
//      insertion is a ballot/popcount rank + one shuffle.  Order is (distance, index)
//      lexicographic => ties resolve to the lowest gallery index, independent of scan order.
#include "common.h"

#include <algorithm>

namespace vtcgemm { int num_cus(); }   // gemm.hip

namespace {

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ x, float *__restrict__ out, int n, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float *xr = x + (size_t)r * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c] * xr[c];
  s = wave_sum(s);
  if (lane == 0) out[r] = s;
}

// out[r] = [hi | lo | hi] (query side, side = 0) or [hi | hi | lo] (gallery side, side = 1); parts = 1 => [hi]
__global__ __launch_bounds__(256) void split_bf16_kernel(const float *__restrict__ x, bf16_t *__restrict__ out, int n, int d,
                                                         int parts, int side) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * d) return;
  const size_t r = i / d;
  const int c = (int)(i - r * d);
  const float v = x[i];
  const bf16_t hi = f2bf(v);
  bf16_t *o = out + r * (size_t)(parts * d);
  o[c] = hi;
  if (parts == 3) {
    const bf16_t lo = f2bf(v - bf2f(hi));
    o[d + c] = side == 0 ? lo : hi;
    o[2 * d + c] = side == 0 ? hi : lo;
  }
}

// ---- fused prologue of the block-minima sweep (round 4): one launch for BOTH embedding sets ---------------------------------
// Per row x (one wave): |x|^2 (fp32), the bf16 operand row x~ = rne(x) (xb == NULL: statistics only), and the two numbers the
// certificate's error bound is made of -- |x~| and |e| with e = x - x~, the row's ACTUAL rounding error (exact in fp32: x and x~
// agree in sign, exponent and the top mantissa bits, so the difference is representable) -- plus the running maxima of all three
// over the set (per-block maxima, folded by sweep_prep_max_kernel).
//   | q~.g~ - q.g | = | q~.e_g + e_q.g~ + e_q.e_g |  <=  |q~||e_g| + |e_q||g~| + |e_q||e_g|          (Cauchy-Schwarz)
// With |e| <= 2^-8 |x| this is the worst-case bound of rounds 2-3 (2u + u^2)|q||g|; on real embeddings |e| ~ 0.4 x 2^-8 |x|, so
// the bound is ~0.45 x that and the candidate sets shrink accordingly -- and rows that do sit on bf16 midpoints (the adversarial test) still get the
// full bound, because |e| is measured, not assumed.  Both norms carry a 1e-4 relative slack for their fp32 summation.
struct PrepSide {
  const float *x;
  bf16_t *xb;
  float *n2;
  float2 *st;      // (|x~|, |e|) per row
  float *mx;       // [0] max |x|^2, [1] max |x~|, [2] max |e|   (written by sweep_prep_max_kernel)
  int n;
};
constexpr int PREP_RPW = 4;      // rows per wave
// part[block][8]: the block's maxima {A: |x|^2, |x~|, |e|, -; B: ...} -- no atomics (20 000 rows x 3 atomic maxima on two sets of
// three words took 0.5 ms); sweep_prep_max_kernel folds the blocks' rows
__global__ __launch_bounds__(256) void sweep_prep_kernel(PrepSide A, PrepSide B, int d, float *__restrict__ part) {
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < PREP_RPW; ++i) {
    int r = (blockIdx.x * 4 + wave) * PREP_RPW + i;
    if (r >= A.n + B.n) break;                   // wave-uniform
    const bool second = r >= A.n;
    const PrepSide &S = second ? B : A;
    if (second) r -= A.n;
    const float *xr = S.x + (size_t)r * d;
    float n2 = 0.f, nt2 = 0.f, e2 = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(xr + c);
      const float f[4] = {v.x, v.y, v.z, v.w};
      unsigned short h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = f2bf(f[j]);
        const float t = bf2f(h[j]), e = f[j] - t;
        n2 = fmaf(f[j], f[j], n2);
        nt2 = fmaf(t, t, nt2);
        e2 = fmaf(e, e, e2);
      }
      if (S.xb) *reinterpret_cast<uint2 *>(S.xb + (size_t)r * d + c) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    }
    n2 = wave_sum(n2); nt2 = wave_sum(nt2); e2 = wave_sum(e2);
    // (+ d x FLT_MIN under the roots: a component below ~1e-19 squares to a flushed zero, and up to d such terms may be missing from
    //  either sum -- ADVICE r4; on unit-norm rows the addend is 6e-36 beside 1e-6 .. 1 and changes no bit)
    const float nt = sqrtf(nt2 + (float)d * 1.1754944e-38f) * 1.0001f, e = sqrtf(e2 + (float)d * 1.1754944e-38f) * 1.0001f;
    if (lane == 0) {
      S.n2[r] = n2;
      S.st[r] = make_float2(nt, e);
    }
    const int o = second ? 4 : 0;
    mx[o] = fmaxf(mx[o], n2); mx[o + 1] = fmaxf(mx[o + 1], nt); mx[o + 2] = fmaxf(mx[o + 2], e);
  }
  if (lane < 8) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = lane == k ? mx[k] : v;
    red[wave][lane] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) part[(size_t)blockIdx.x * 8 + threadIdx.x] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}
__global__ __launch_bounds__(256) void sweep_prep_max_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ mxA, float *__restrict__ mxB,
                                                             int *zero_a, int *zero_b) {
  __shared__ float red[32][8];
  if (threadIdx.x == 0) {          // (the recall-rank path's two fallback counters: saves a fill launch)
    if (zero_a) *zero_a = 0;
    if (zero_b) *zero_b = 0;
  }
  const int k = threadIdx.x & 7, sl = threadIdx.x >> 3;
  float m = 0.f;
  for (int b = sl; b < nblocks; b += 32) m = fmaxf(m, part[(size_t)b * 8 + k]);
  red[sl][k] = m;
  __syncthreads();
  if (threadIdx.x < 8) {
    float v = 0.f;
    for (int i = 0; i < 32; ++i) v = fmaxf(v, red[i][threadIdx.x]);
    if (threadIdx.x < 3) mxA[threadIdx.x] = v;
    else if (threadIdx.x >= 4 && threadIdx.x < 7) mxB[threadIdx.x - 4] = v;
  }
}
static int prep_blocks(int rows) { return cdiv(rows, 4 * PREP_RPW); }
static void launch_sweep_prep(const PrepSide &A, const PrepSide &B, int d, float *part, hipStream_t stream, int *zero_a = nullptr, int *zero_b = nullptr) {
  const int nb = prep_blocks(A.n + B.n);
  hipLaunchKernelGGL(sweep_prep_kernel, dim3(nb), dim3(256), 0, stream, A, B, d, part);
  hipLaunchKernelGGL(sweep_prep_max_kernel, dim3(1), dim3(256), 0, stream, part, nb, A.mx, B.mx, zero_a, zero_b);
}

// Wave-wide sorted best list, one entry per lane (lane i = i-th best), order = (distance, index).
struct WaveList {
  float bd;
  int bi;
  float tau;     // distance of entry depth-1 (wave-uniform)
  int tau_i;
  __device__ __forceinline__ void init() { bd = INFINITY; bi = 0x7fffffff; tau = INFINITY; tau_i = 0x7fffffff; }
  // offer one candidate per lane (pass = lane has a candidate that beats the current depth-th entry).
  // Insertions are rare after warm-up but serial: broadcasts are v_readlane (the lane index is
  // wave-uniform), the one-lane shift a ds_bpermute.
  __device__ __forceinline__ void offer(float v, int idx, bool pass, int lane, int depth) {
    unsigned long long mask = __ballot(pass);
    while (mask) {
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      const float cv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
      const int ci = __builtin_amdgcn_readlane(idx, l);
      if (cv < tau || (cv == tau && ci < tau_i)) {       // wave-uniform: the list may have tightened meanwhile
        const bool less = (bd < cv) || (bd == cv && bi < ci);
        const int pos = __popcll(__ballot(less));
        // shift by one lane: DPP wave_shr:1 (a VALU move, no LDS round trip as __shfl_up's ds_bpermute)
        const float ud = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, bd), 0x138, 0xf, 0xf, false));
        const int ui = __builtin_amdgcn_update_dpp(0, bi, 0x138, 0xf, 0xf, false);
        if (lane > pos) { bd = ud; bi = ui; }
        if (lane == pos) { bd = cv; bi = ci; }
        tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bd), depth - 1));
        tau_i = __builtin_amdgcn_readlane(bi, depth - 1);
      }
    }
  }
  __device__ __forceinline__ bool beats(float v, int idx) const { return v < tau || (v == tau && idx < tau_i); }
};

// One wave per (row, column segment): streams its segment 1024 columns per step -- four 16-byte loads per lane,
// the next TWO steps' loads already in flight while this step is screened (8 KiB per wave outstanding; with the ~2 us
// loaded latency of this path two 16-byte loads per wave and 13 waves per CU left the kernel latency-bound at
// 0.8-1.2 TB/s).  Screening is one min over the lane's 16 values against the wave-uniform k-th best: after
// warm-up almost every step is 4 loads + 15 v_min + 1 ballot.  Segments exist only to put enough waves on the
// chip when a block has few rows; S == 1 writes the final ids/dists directly.
__global__ __launch_bounds__(256) void row_topk_kernel(const float *__restrict__ dist, int ld, int n_rows, int n_cols, int depth,
                                                       int S, int seg_cols, int64_t *__restrict__ ids, float *__restrict__ dists,
                                                       size_t out_row0, float *__restrict__ part_d, int *__restrict__ part_i) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_rows * S) return;
  const int r = w / S, seg = w - r * S;
  const float *row = dist + (size_t)r * ld;
  const int c_lo = seg * seg_cols, c_hi = min(n_cols, c_lo + seg_cols);
  WaveList wl;
  wl.init();
  const bool vec = (ld & 3) == 0;
  // lane's 4 x 4 columns of the step starting at `base`: base + 256 h + 4 lane + e
  auto load_step = [&](int base, float (&v)[16]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int c = base + 256 * h + lane * 4;
      if (vec && c + 3 < c_hi) {
        typedef float v4f_t __attribute__((ext_vector_type(4)));
        const v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_t *>(row + c));   // read once
        v[4 * h] = t.x; v[4 * h + 1] = t.y; v[4 * h + 2] = t.z; v[4 * h + 3] = t.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = c + e < c_hi ? row[c + e] : INFINITY;
      }
    }
  };
  float cur[16], nxt[16], nx2[16];
  load_step(c_lo, cur);
  if (c_lo + 1024 < c_hi) load_step(c_lo + 1024, nxt);
  for (int base = c_lo; base < c_hi; base += 1024) {
    if (base + 2048 < c_hi) load_step(base + 2048, nx2);     // two steps (8 KiB per wave) in flight behind this one
    float m = cur[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) m = fminf(m, cur[q]);
    // m <= tau is necessary for any of the 16 to beat (tau, tau_i); columns past c_hi hold +inf and never insert
    if (__ballot(m <= wl.tau) != 0ull) {
      if (base == c_lo) {
        // Warm-up: with an empty list every value "beats" it and 1024 candidates would be offered one by one.
        // Offer each lane's smallest first (64 candidates): the list's k-th best is then already within a few
        // ranks of the step's true k-th best and the remaining 960 are screened against it.
        int qm = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) qm = cur[q] < cur[qm] ? q : qm;     // first minimum = lowest column
        float vm = cur[0];
#pragma unroll
        for (int q = 1; q < 16; ++q) vm = q == qm ? cur[q] : vm;
        const int im = base + 256 * (qm >> 2) + lane * 4 + (qm & 3);
        if (base + 1024 <= c_hi) {
          // a full first step: the 64 lane minima ARE the initial list, sorted by rank counting + one forward
          // permute (as col_topk_kernel) instead of 64 serial insertions
          int rank = 0;
#pragma unroll 4
          for (int l = 0; l < 64; ++l) {
            const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vm), l));
            const int oi = __builtin_amdgcn_readlane(im, l);
            rank += (ov < vm || (ov == vm && oi < im)) ? 1 : 0;
          }
          wl.bd = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(rank << 2, __builtin_bit_cast(int, vm)));
          wl.bi = __builtin_amdgcn_ds_permute(rank << 2, im);
          wl.tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl.bd), depth - 1));
          wl.tau_i = __builtin_amdgcn_readlane(wl.bi, depth - 1);
        } else {
          wl.offer(vm, im, im < c_hi, lane, depth);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int idx = base + 256 * (q >> 2) + lane * 4 + (q & 3);
          wl.offer(cur[q], idx, q != qm && idx < c_hi && wl.beats(cur[q], idx), lane, depth);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int idx = base + 256 * (q >> 2) + lane * 4 + (q & 3);
          wl.offer(cur[q], idx, idx < c_hi && wl.beats(cur[q], idx), lane, depth);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) { cur[q] = nxt[q]; nxt[q] = nx2[q]; }
  }
  if (lane < depth) {
    if (S == 1) {
      ids[(out_row0 + r) * depth + lane] = wl.bi == 0x7fffffff ? -1 : (int64_t)wl.bi;
      if (dists) dists[(out_row0 + r) * depth + lane] = wl.bd;
    } else {
      part_d[(size_t)w * depth + lane] = wl.bd;
      part_i[(size_t)w * depth + lane] = wl.bi;
    }
  }
}

// Column direction of the same matrix (the second retrieval direction needs D^T, so it is read off the block
// that the first direction's GEMM has just written instead of running a second GEMM): one workgroup per
// (strip of 64 columns, row segment).  The segment's rows stream through LDS in 64 x 64 tiles (coalesced 256-byte row
// pieces in, 16-byte LDS accesses both ways: row stride 68 floats); wave w owns columns 16w .. 16w+15 of the strip and
// keeps one WaveList per column (2 VGPRs each).  The first tile of a segment initialises a list by rank-counting
// its 64 values (64 serial insertions per column otherwise).  ids are row_id0 + row; partial lists live in
// part[(col * S_total + seg0 + seg) * depth ..], are CARRIED from one block of rows to the next (carry != 0) and are
// merged over segments by topk_merge_kernel.
__global__ __launch_bounds__(256, 4) void col_topk_kernel(const float *__restrict__ dist, int ld, int n_rows, int n_cols, int depth,
                                                       int row_id0, int S, int seg_rows, int S_total, int seg0, int carry,
                                                       float *__restrict__ part_d, int *__restrict__ part_i) {
  __shared__ __attribute__((aligned(16))) float tile[64 * 68];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int strip = blockIdx.x / S, seg = blockIdx.x - strip * S;
  const int c0 = strip * 64;
  const int r_lo = seg * seg_rows, r_hi = min(n_rows, r_lo + seg_rows);
  if (r_lo >= r_hi) {            // empty segment (uniform): its partial lists stay as they are / "invalid"
    if (carry) return;
    for (int cc = 0; cc < 16; ++cc) {
      const int col = c0 + 16 * w + cc;
      if (col < n_cols && lane < depth) {
        part_d[((size_t)col * S_total + seg0 + seg) * depth + lane] = INFINITY;
        part_i[((size_t)col * S_total + seg0 + seg) * depth + lane] = 0x7fffffff;
      }
    }
    return;
  }
  WaveList wl[16];
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    wl[cc].init();
    const int col = c0 + 16 * w + cc;
    if (carry && col < n_cols) {
      // the list of this (column, segment) continues from the previous block of rows: its k-th best is already a
      // tight threshold, so later blocks insert ~k ln(rows_after / rows_before) times instead of ~k ln(rows / 64)
      if (lane < depth) {
        wl[cc].bd = part_d[((size_t)col * S_total + seg0 + seg) * depth + lane];
        wl[cc].bi = part_i[((size_t)col * S_total + seg0 + seg) * depth + lane];
      }
      wl[cc].tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl[cc].bd), depth - 1));
      wl[cc].tau_i = __builtin_amdgcn_readlane(wl[cc].bi, depth - 1);
    }
    if (col >= n_cols) wl[cc].tau = -INFINITY;                  // columns past the matrix never offer anything
  }
  const int lr = t >> 4, lc = (t & 15) * 4;
  const bool vec = (ld & 3) == 0 && c0 + 64 <= n_cols;      // uniform: whole strip inside the matrix, rows 16-byte aligned
  auto load_tile = [&](int r_base, float4 (&pre)[4]) {
    if (r_base >= r_hi) return;                              // uniform
    if (vec) {
      // branch-free: rows past the segment re-read its last row and are replaced by +inf, so the four loads of a
      // thread are all in flight together
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int r = r_base + lr + 16 * h;
        pre[h] = *reinterpret_cast<const float4 *>(dist + (size_t)min(r, r_hi - 1) * ld + c0 + lc);
      }
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int r = r_base + lr + 16 * h;
        float4 v = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
        if (r < r_hi) {
          const float *src = dist + (size_t)r * ld + c0 + lc;
          if (c0 + lc < n_cols) v.x = src[0];
          if (c0 + lc + 1 < n_cols) v.y = src[1];
          if (c0 + lc + 2 < n_cols) v.z = src[2];
          if (c0 + lc + 3 < n_cols) v.w = src[3];
        }
        pre[h] = v;
      }
    }
  };
  // one 64-row step: registers -> LDS, refill the registers with the next tile, screen this tile from LDS
  auto step = [&](int r_base, float4 (&pre)[4]) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const bool live = r_base + lr + 16 * h < r_hi;
      *reinterpret_cast<float4 *>(tile + (lr + 16 * h) * 68 + lc) =
          make_float4(live ? pre[h].x : INFINITY, live ? pre[h].y : INFINITY, live ? pre[h].z : INFINITY, live ? pre[h].w : INFINITY);
    }
    __syncthreads();
    load_tile(r_base + 64, pre);                             // the next tile is in flight while this one is screened
    const bool valid = r_base + lane < r_hi;
    const int idx = valid ? row_id0 + r_base + lane : 0x7fffffff;
    const bool sort_init = !carry && r_base == r_lo && r_lo + 64 <= r_hi;   // a full first tile of a fresh list (uniform)
    float vv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t4 = *reinterpret_cast<const float4 *>(tile + lane * 68 + 16 * w + 4 * q);
      vv[4 * q] = t4.x; vv[4 * q + 1] = t4.y; vv[4 * q + 2] = t4.z; vv[4 * q + 3] = t4.w;
    }
    if (sort_init) {
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) {
        const float v = vv[cc];
        int rank = 0;
#pragma unroll 4
        for (int l = 0; l < 64; ++l) {
          const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
          const int oi = __builtin_amdgcn_readlane(idx, l);
          rank += (ov < v || (ov == v && oi < idx)) ? 1 : 0;
        }
        // ranks are a permutation (indices are distinct): forward permute = scatter to lane `rank`
        wl[cc].bd = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(rank << 2, __builtin_bit_cast(int, v)));
        wl[cc].bi = __builtin_amdgcn_ds_permute(rank << 2, idx);
        wl[cc].tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl[cc].bd), depth - 1));
        wl[cc].tau_i = __builtin_amdgcn_readlane(wl[cc].bi, depth - 1);
        if (c0 + 16 * w + cc >= n_cols) wl[cc].tau = -INFINITY;   // columns past the matrix never offer anything
      }
    } else {
      // hot path: one compare per column, ORed; the per-column control flow runs only when some lane of the wave
      // has a candidate in some column
      bool anyp = false;
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) anyp |= vv[cc] <= wl[cc].tau;
      if (__ballot(valid && anyp) != 0ull) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
          wl[cc].offer(vv[cc], idx, valid && vv[cc] <= wl[cc].tau, lane, depth);   // offer() re-checks (tau, tau_i) exactly
      }
    }
    __syncthreads();
  };
  float4 pre[4];
  load_tile(r_lo, pre);
  for (int r_base = r_lo; r_base < r_hi; r_base += 64) step(r_base, pre);
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const int col = c0 + 16 * w + cc;
    if (col < n_cols && lane < depth) {
      part_d[((size_t)col * S_total + seg0 + seg) * depth + lane] = wl[cc].bd;
      part_i[((size_t)col * S_total + seg0 + seg) * depth + lane] = wl[cc].bi;
    }
  }
}

// One wave per row: merge the S partial lists of the row.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float *__restrict__ part_d, const int *__restrict__ part_i, int n_rows,
                                                         int S, int depth, int64_t *__restrict__ ids, float *__restrict__ dists,
                                                         size_t out_row0) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  WaveList wl;
  wl.init();
  const int total = S * depth;
  for (int base = 0; base < total; base += 64) {
    const int k = base + lane;
    const float v = k < total ? part_d[(size_t)r * total + k] : INFINITY;
    const int idx = k < total ? part_i[(size_t)r * total + k] : 0x7fffffff;
    wl.offer(v, idx, k < total && idx != 0x7fffffff && wl.beats(v, idx), lane, depth);
  }
  if (lane < depth) {
    ids[(out_row0 + r) * depth + lane] = wl.bi == 0x7fffffff ? -1 : (int64_t)wl.bi;
    if (dists) dists[(out_row0 + r) * depth + lane] = wl.bd;
  }
}

// ---- block-minima path: certified candidate sets from the per-block keys of the EPI_L2MIN epilogue ----------------------
// keys[4][nblk][R]: for owner r (a query row; a gallery column in the transposed direction) and block blk of BW gallery
// entries, the three smallest distance keys (the POOL) and the fourth smallest (a bound on everything else in the block).
// With eps_r >= |approx - exact| for every entry of the row:
//   u      = the depth-th smallest pool key of the row     (depth DISTINCT entries lie at or below it, so the exact
//                                                           depth-th distance t satisfies t <= u + eps)
//   theta  = u + 2 eps                                      (every true top-depth entry has approx <= t + eps <= theta)
//   C_r    = all pool entries with key <= theta             (the candidate set handed to the fp64 re-rank)
//   proof of completeness: if the smallest fourth-in-block key of the row is > theta, no entry outside the pool can have
//   approx <= theta, hence C_r contains every entry with approx <= theta, hence the true top-depth.  Otherwise (or if
//   |C_r| > 64) cand_n[r] = -1 and the row is recomputed by fp64 brute force.
// One workgroup per 64 owners; blocks of 64 x 64 keys are transposed through LDS (coalesced 256-byte reads in, one
// owner's 64 keys per wave read out).  Pass 1 reads plane 0 only (u is taken over block minima: still depth distinct
// entries, a valid if slightly larger u); pass 2 reads all four planes and emits C_r with ballot compaction.
// Key layout: keys[src][plane][block-of-src][owner] with `nbs` blocks per source (one source: the planes of a local GEMM,
// nbs == nblk; several: the column planes every rank sent for this rank's columns, stacked in rank order) -- block `blk`
// holds entries src_base[blk / nbs] + (blk % nbs) * bw + (key & 127) of the other side (src_base == NULL: 0).
// OW owners per workgroup: 64, or 32 when 64 would leave CUs without a workgroup (10k x 10k: 157 workgroups of 64)
// (round 4: ONE launch serves both directions of the one-matrix sweep -- workgroups [0, nblocks_a) take problem A, the rest B;
// nblocks_a == gridDim.x: a single problem.  The same holds for exact_rerank_kernel and block_rescan_kernel below.)
struct MinselArgs {
  const unsigned *keys;
  int R, nblk, bw, nbs;
  const int *src_base;
  int depth;
  const float *own_norm;
  const float2 *own_st;
  const float *other_max;
  float kappa;
  int64_t *cand;
  int *cand_n;
  float *theta_out;
};
template <int OW>
__global__ __launch_bounds__(256) void minsel_kernel(const MinselArgs PA, const MinselArgs PB, int nblocks_a) {
  const bool second = (int)blockIdx.x >= nblocks_a;
  const MinselArgs &P = second ? PB : PA;
  const int bid = second ? (int)blockIdx.x - nblocks_a : (int)blockIdx.x;
  const unsigned *__restrict__ keys = P.keys;
  const int R = P.R, nblk = P.nblk, bw = P.bw, nbs = P.nbs, depth = P.depth;
  const int *__restrict__ src_base = P.src_base;
  const float *__restrict__ own_norm = P.own_norm;
  const float2 *__restrict__ own_st = P.own_st;
  const float *__restrict__ other_max = P.other_max;
  const float kappa = P.kappa;
  int64_t *__restrict__ cand = P.cand;
  int *__restrict__ cand_n = P.cand_n;
  float *__restrict__ theta_out = P.theta_out;
  constexpr int TG = OW / 4, NSUB = 256 / TG, NI = 64 / NSUB;      // vector loads: TG threads along the owners, NSUB block slices
  constexpr int OPW = OW / 4;                                       // owners per wave
  constexpr int SROWS = 256 / OW, NQ = 64 / SROWS;                  // scalar loads: a thread per owner, SROWS block slices
  static_assert(OW == 64 || OW == 32, "owners per workgroup");
  __shared__ unsigned tl[3][64 * 65];
  __shared__ float bmin[32][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r0 = bid * OW;
  const size_t plane = (size_t)nbs * R;                       // one plane of one source
  auto blk_off = [&](int blk) -> size_t {                      // offset of (plane 0, blk, owner 0)
    const int src = blk / nbs;
    return ((size_t)src * (L2MIN_PLANES - 1) * nbs + blk) * R;   // = ((src * 4) * nbs + blk % nbs) * R
  };
  const int so = t % OW, sb = t / OW;                           // scalar loads: this thread's owner and block slice
  const bool rv = r0 + so < R;
  const float INF = __builtin_bit_cast(float, 0x7F800000u);
  // ---- pass 1: per owner, the two smallest block minima each lane has seen ----
  float m1[OPW], m2[OPW];
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) m1[cc] = m2[cc] = INF;
  // tile loads: 16-byte loads (four consecutive owners per thread, all of a chunk's loads in flight at once) when the
  // planes' rows are 16-byte aligned, else one key per thread
  const bool vec = (R & 3) == 0;
  const int og = 4 * (t % TG), sub = t / TG;
  float bq[4] = {INF, INF, INF, INF};
  auto load_chunk = [&](int c0, int npl) __attribute__((always_inline)) {
    if (vec) {
      uint4 v[NI][4];
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int blk = c0 + sub + NSUB * it;
        const bool ok = blk < nblk && r0 + og < R;
        const size_t o = (ok ? blk_off(blk) : 0) + r0 + og;
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
          if (pl < npl) v[it][pl] = ok ? *reinterpret_cast<const uint4 *>(keys + pl * plane + o) : make_uint4(0x7F800000u, 0x7F800000u, 0x7F800000u, 0x7F800000u);
      }
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int bl = sub + NSUB * it;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          if (pl < npl) {
            tl[pl][bl * 65 + og + 0] = v[it][pl].x; tl[pl][bl * 65 + og + 1] = v[it][pl].y;
            tl[pl][bl * 65 + og + 2] = v[it][pl].z; tl[pl][bl * 65 + og + 3] = v[it][pl].w;
          }
        if (npl == 4) {     // fourth-in-block keys: running min per owner, reduced over the threads at the end
          bq[0] = fminf(bq[0], __uint_as_float(v[it][3].x)); bq[1] = fminf(bq[1], __uint_as_float(v[it][3].y));
          bq[2] = fminf(bq[2], __uint_as_float(v[it][3].z)); bq[3] = fminf(bq[3], __uint_as_float(v[it][3].w));
        }
      }
    } else {
#pragma unroll 4
      for (int q = 0; q < NQ; ++q) {
        const int bl = sb + SROWS * q, blk = c0 + bl;
        const bool ok = blk < nblk && rv;
        const size_t o = (ok ? blk_off(blk) : 0) + r0 + so;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          if (pl < npl) tl[pl][bl * 65 + so] = ok ? keys[pl * plane + o] : 0x7F800000u;
        if (npl == 4 && ok) bq[0] = fminf(bq[0], __uint_as_float(keys[3 * plane + o]));
      }
    }
  };
  for (int c0 = 0; c0 < nblk; c0 += 64) {
    load_chunk(c0, 1);
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < OPW; ++cc) {
      const float k = __uint_as_float(tl[0][lane * 65 + OPW * w + cc]);
      m2[cc] = fminf(m2[cc], fmaxf(m1[cc], k));
      m1[cc] = fminf(m1[cc], k);
    }
    __syncthreads();
  }
  // depth-th smallest of the wave's 128 values per owner: the 64 lane minima sorted by rank counting, then the few second
  // minima that beat the running depth-th
  float theta[OPW];
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) {
    const float v = m1[cc];
    int rank = 0;
#pragma unroll 4
    for (int l = 0; l < 64; ++l) {
      const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
      rank += (ov < v || (ov == v && l < lane)) ? 1 : 0;
    }
    WaveList wl;
    wl.bd = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(rank << 2, __builtin_bit_cast(int, v)));
    wl.bi = __builtin_amdgcn_ds_permute(rank << 2, lane);
    wl.tau = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl.bd), depth - 1));
    wl.tau_i = __builtin_amdgcn_readlane(wl.bi, depth - 1);
    wl.offer(m2[cc], 64 + lane, wl.beats(m2[cc], 64 + lane), lane, depth);
    const float u = wl.tau;
    const int r = min(r0 + OPW * w + cc, R - 1);
    // |approx - exact| of any entry of this owner's row (sweep_prep_kernel: measured rounding errors; other_max = the other
    // side's {max |x|^2, max |x~|, max |e|}) + kappa x the fp32 terms (accumulation, norms, the key's index bits)
    const float2 st = own_st[r];
    const float eps = 1.001f * (2.0f * (st.x * other_max[2] + st.y * other_max[1] + st.y * other_max[2]) + kappa * (own_norm[r] + other_max[0]));
    theta[cc] = u + 2.0f * eps;          // u == +inf (fewer than depth pool entries) keeps theta infinite: everything is a candidate
  }
  // ---- pass 2: every pool entry with key <= theta, and the smallest fourth-in-block key ----
  int cnt[OPW];
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) cnt[cc] = 0;
  for (int c0 = 0; c0 < nblk; c0 += 64) {
    load_chunk(c0, 4);
    __syncthreads();
    const int blk = c0 + lane;
    const int64_t base = blk < nblk ? (int64_t)(src_base ? src_base[blk / nbs] : 0) + (int64_t)(blk % nbs) * bw : 0;
#pragma unroll
    for (int cc = 0; cc < OPW; ++cc) {
      const int o = OPW * w + cc;
      const int r = r0 + o;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const unsigned k = tl[pl][lane * 65 + o];
        const bool pass = __uint_as_float(k & ~127u) <= theta[cc] && k != 0x7F800000u;
        const unsigned long long mask = __ballot(pass);
        if (mask) {                                                       // wave-uniform
          const int pos = cnt[cc] + __popcll(mask & ((1ull << lane) - 1ull));
          if (pass && pos < 64 && r < R) cand[(size_t)r * 64 + pos] = base + (int)(k & 127u);
          cnt[cc] += __popcll(mask);
        }
      }
    }
    __syncthreads();
  }
  // smallest fourth-in-block key per owner: the threads' running minima -> bmin[slice][owner] -> min over the slices
  for (int i = t; i < 32 * 64; i += 256) (&bmin[0][0])[i] = INF;
  __syncthreads();
  if (vec) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bmin[sub][og + e] = bq[e];
  } else {
    bmin[sb][so] = bq[0];
  }
  __syncthreads();
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) {
    const int o = OPW * w + cc, r = r0 + o;
    if (r < R && lane == 0) {
      float bound = INF;
      for (int sl = 0; sl < 32; ++sl) bound = fminf(bound, bmin[sl][o]);
      // strict: an outside entry AT theta could tie with a true neighbour
      cand_n[r] = (cnt[cc] <= 64 && bound > theta[cc]) ? cnt[cc] : -1;
      theta_out[r] = theta[cc];                            // for the uncertified owners' second pass (block_rescan_kernel)
    }
  }
}

static MinselArgs minsel_args(const unsigned *keys, int R, int nblk, int bw, int nbs, const int *src_base, int depth, const float *own_norm,
                              const float2 *own_st, const float *other_max, float kappa, int64_t *cand, int *cand_n, float *theta) {
  return MinselArgs{keys, R, nblk, bw, nbs, src_base, depth, own_norm, own_st, other_max, kappa, cand, cand_n, theta};
}
// one problem (B == nullptr) or both directions in one launch
static void launch_minsel(const MinselArgs &A, const MinselArgs *B, hipStream_t stream) {
  const int total = A.R + (B ? B->R : 0);
  const MinselArgs &Bv = B ? *B : A;
  if (cdiv(total, 64) < vtcgemm::num_cus()) {
    const int na = cdiv(A.R, 32), nb = B ? cdiv(B->R, 32) : 0;
    hipLaunchKernelGGL(minsel_kernel<32>, dim3(na + nb), dim3(256), 0, stream, A, Bv, na);
  } else {
    const int na = cdiv(A.R, 64), nb = B ? cdiv(B->R, 64) : 0;
    hipLaunchKernelGGL(minsel_kernel<64>, dim3(na + nb), dim3(256), 0, stream, A, Bv, na);
  }
}

// ---- VTC_SWEEP_EXACT ------------------------------------------------------------------------------------
// fp64 squared distance sum_k (q_k - g_k)^2 of one (query, gallery) pair by the whole wave (lanes stride over k)
__device__ __forceinline__ double wave_dist64(const float *__restrict__ q, const float *__restrict__ g, int d, int lane) {
  double s = 0.0;
  for (int c = lane * 4; c < d; c += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(q + c), b = *reinterpret_cast<const float4 *>(g + c);
    const double e0 = (double)a.x - (double)b.x, e1 = (double)a.y - (double)b.y, e2 = (double)a.z - (double)b.z,
                 e3 = (double)a.w - (double)b.w;
    s += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);   // xor butterfly: every lane ends with the same bits
  return s;
}

// One wave per query row: fp64 distances of the row's `cdepth` BF16X3 candidates, sorted by (distance, index); the
// first `depth` are the answer IF the candidate list provably contains the true top-`depth`: every true member j
// has approx(j) <= approx_(depth) + 2 eps, so it is on the list when approx_(cdepth) > approx_(depth) + 2 eps, with
// eps = kappa (|q|^2 + max|g|^2) a worst-case bound of the split-bf16 distance error (dropped lo.lo products, the
// two bf16 roundings of each operand, fp32 accumulation of 3 D products).  Otherwise the row is flagged.
struct RerankArgs {
  const float *queries, *gallery;
  int nq, ng, d;
  const int64_t *cand;
  const float *cand_d;
  int cdepth, depth;
  const float *qn, *gmax;
  float kappa;
  int64_t *ids;
  float *dists;
  int *flags;          // flags[0] = count, flags[1..] = rows
  const int *cand_n;   // block-minima path: the row's list holds cand_n[r] entries and is ALREADY certified complete (< 0: it is not); else nullptr
};
__global__ __launch_bounds__(256) void exact_rerank_kernel(const RerankArgs PA, const RerankArgs PB, int nblocks_a) {
  const bool second = (int)blockIdx.x >= nblocks_a;
  const RerankArgs &P = second ? PB : PA;
  const int bid = second ? (int)blockIdx.x - nblocks_a : (int)blockIdx.x;
  const float *__restrict__ queries = P.queries, *__restrict__ gallery = P.gallery;
  const int nq = P.nq, ng = P.ng, d = P.d, cdepth = P.cdepth, depth = P.depth;
  const int64_t *__restrict__ cand = P.cand;
  const float *__restrict__ cand_d = P.cand_d, *__restrict__ qn = P.qn, *__restrict__ gmax = P.gmax;
  const float kappa = P.kappa;
  int64_t *__restrict__ ids = P.ids;
  float *__restrict__ dists = P.dists;
  int *__restrict__ flags = P.flags;
  const int *__restrict__ cand_n = P.cand_n;
  const int lane = threadIdx.x & 63;
  const int r = bid * 4 + (threadIdx.x >> 6);
  if (r >= nq) return;
  const float *q = queries + (size_t)r * d;
  const int n_list = cand_n ? cand_n[r] : cdepth;
  const int64_t my = lane < n_list ? cand[(size_t)r * cdepth + lane] : -1;
  double md = INFINITY;
  // Eight candidates at a time: their row pieces are all requested before the first is used and the eight xor
  // butterflies interleave (one candidate at a time the kernel was a chain of 32 dependent gather latencies).
  // Per candidate the arithmetic and its order are exactly wave_dist64's, so the fallback kernel forms the same doubles.
  const int n_loop = cand_n ? (n_list > 0 ? n_list : 0) : cdepth;      // wave-uniform
  constexpr int NB = 8;                                                // candidates per step
  for (int c0 = 0; c0 < n_loop; c0 += NB) {
    long long jj[NB];
    double sacc[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      jj[u] = c0 + u < n_loop ? __shfl((long long)my, c0 + u, 64) : -1;      // wave-uniform
      sacc[u] = 0.0;
    }
    for (int c = lane * 4; c < d; c += 256) {
      const float4 a = *reinterpret_cast<const float4 *>(q + c);
      float4 b[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u)
        b[u] = *reinterpret_cast<const float4 *>(gallery + (size_t)(jj[u] < 0 ? 0 : jj[u]) * d + c);
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const double e0 = (double)a.x - (double)b[u].x, e1 = (double)a.y - (double)b[u].y, e2 = (double)a.z - (double)b[u].z,
                     e3 = (double)a.w - (double)b[u].w;
        sacc[u] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < NB; ++u) sacc[u] += __shfl_xor(sacc[u], o, 64);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (jj[u] >= 0 && lane == c0 + u) md = sacc[u];
  }
  // rank among the candidates by (fp64 distance, index); absent slots sort last
  int rank = 0;
  for (int c = 0; c < n_loop; ++c) {
    const double od = __shfl(md, c, 64);
    const long long oi = __shfl((long long)my, c, 64);
    if (oi >= 0 && (od < md || (od == md && oi < (long long)my))) ++rank;
  }
  if (lane < cdepth && my >= 0 && rank < depth) {
    ids[(size_t)r * depth + rank] = my;
    if (dists) dists[(size_t)r * depth + rank] = (float)md;
  }
  const int n_valid = __popcll(__ballot(my >= 0));
  if (lane == 0) {
    bool sure = n_valid >= ng;                            // the list IS the gallery
    if (cand_n) {
      sure = sure || n_valid >= depth;                    // certified by minsel_kernel (n_list < 0 -> n_valid == 0)
    } else if (!sure && n_valid >= depth) {
      const float eps = kappa * (qn[r] + *gmax);
      // everything outside the list is at least as far (approx) as its last entry
      const float outside = n_valid >= cdepth ? cand_d[(size_t)r * cdepth + cdepth - 1] : INFINITY;
      sure = outside > cand_d[(size_t)r * cdepth + depth - 1] + 2.0f * eps;
    }
    if (!sure) flags[1 + atomicAdd(flags, 1)] = r;
  }
}

// Flagged rows only (dense near-ties: duplicates in the gallery): fp64 brute force over the whole gallery, one
// workgroup per row, a sorted (distance, index) list per wave (one entry per lane), merged by wave 0.
__global__ __launch_bounds__(256) void exact_fallback_kernel(const float *__restrict__ queries, const float *__restrict__ gallery, int ng,
                                                             int d, int depth, const int *__restrict__ flags, int64_t *__restrict__ ids,
                                                             float *__restrict__ dists, int f_first) {
  __shared__ double sd[4][64];
  __shared__ int si[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_flagged = flags[0];
  for (int f = f_first + blockIdx.x; f < n_flagged; f += gridDim.x) {   // uniform for the workgroup; normally zero trips
  const int r = flags[1 + f];
  const float *q = queries + (size_t)r * d;
  double bd = INFINITY;
  int bi = 0x7fffffff;
  auto offer = [&](double cv, int ci) {                   // wave-uniform candidate
    const double tau = __shfl(bd, depth - 1, 64);
    const int tau_i = __shfl(bi, depth - 1, 64);
    if (!(cv < tau || (cv == tau && ci < tau_i))) return;
    const bool less = bd < cv || (bd == cv && bi < ci);
    const int pos = __popcll(__ballot(less));
    const double ud = __shfl_up(bd, 1, 64);
    const int ui = __shfl_up(bi, 1, 64);
    if (lane > pos) { bd = ud; bi = ui; }
    if (lane == pos) { bd = cv; bi = ci; }
  };
  for (int j = wave; j < ng; j += 4) offer(wave_dist64(q, gallery + (size_t)j * d, d, lane), j);
  sd[wave][lane] = bd;
  si[wave][lane] = bi;
  __syncthreads();
  if (wave == 0) {
    for (int w = 1; w < 4; ++w)
      for (int e = 0; e < depth; ++e)
        if (si[w][e] != 0x7fffffff) offer(sd[w][e], si[w][e]);
    if (lane < depth) {
      ids[(size_t)r * depth + lane] = bi == 0x7fffffff ? -1 : (int64_t)bi;
      if (dists) dists[(size_t)r * depth + lane] = (float)bd;
    }
  }
  __syncthreads();                                         // sd / si are reused by the next flagged row
  }
}

// The same brute force spread over FB_CHUNKS workgroups per flagged row (the first FB_ROWS flagged rows; a row is a chain of
// dependent gather latencies, so one workgroup per row takes ~10 ms at 50k): workgroup (x, y) scans gallery slice y for the
// flagged rows x, x + gridDim.x, ... and leaves its slice's best `depth` in part_*; fallback_merge_kernel merges the slices.
constexpr int FB_CHUNKS = 512, FB_ROWS = 256, FB_SLOT = 32;   // FB_SLOT >= the largest depth of the block-minima path
__global__ __launch_bounds__(256) void exact_fallback_chunk_kernel(const float *__restrict__ queries, const float *__restrict__ gallery, int ng,
                                                                   int d, int depth, const int *__restrict__ flags, double *__restrict__ part_d,
                                                                   int *__restrict__ part_i) {
  __shared__ double sd[4][64];
  __shared__ int si[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_flagged = min(flags[0], FB_ROWS);
  const int chunk = blockIdx.y;
  const int per = (ng + FB_CHUNKS - 1) / FB_CHUNKS;
  const int j_lo = chunk * per, j_hi = min(ng, j_lo + per);
  for (int f = blockIdx.x; f < n_flagged; f += gridDim.x) {   // uniform for the workgroup; normally zero trips
    const int r = flags[1 + f];
    const float *q = queries + (size_t)r * d;
    double bd = INFINITY;
    int bi = 0x7fffffff;
    auto offer = [&](double cv, int ci) {                   // wave-uniform candidate
      const double tau = __shfl(bd, depth - 1, 64);
      const int tau_i = __shfl(bi, depth - 1, 64);
      if (!(cv < tau || (cv == tau && ci < tau_i))) return;
      const bool less = bd < cv || (bd == cv && bi < ci);
      const int pos = __popcll(__ballot(less));
      const double ud = __shfl_up(bd, 1, 64);
      const int ui = __shfl_up(bi, 1, 64);
      if (lane > pos) { bd = ud; bi = ui; }
      if (lane == pos) { bd = cv; bi = ci; }
    };
    // four gallery rows per step: their loads are all in flight before the first butterfly (same arithmetic and order as
    // wave_dist64 per row, so a distance has the same bits whichever kernel forms it)
    for (int j0 = j_lo + 4 * wave; j0 < j_hi; j0 += 16) {
      double sacc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int c = lane * 4; c < d; c += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(q + c);
        float4 b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const float4 *>(gallery + (size_t)min(j0 + u, j_hi - 1) * d + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double e0 = (double)a.x - (double)b[u].x, e1 = (double)a.y - (double)b[u].y, e2 = (double)a.z - (double)b[u].z,
                       e3 = (double)a.w - (double)b[u].w;
          sacc[u] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) sacc[u] += __shfl_xor(sacc[u], o, 64);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < j_hi) offer(sacc[u], j0 + u);
    }
    sd[wave][lane] = bd;
    si[wave][lane] = bi;
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < 4; ++w)
        for (int e = 0; e < depth; ++e)
          if (si[w][e] != 0x7fffffff) offer(sd[w][e], si[w][e]);
      if (lane < depth) {
        part_d[((size_t)f * FB_CHUNKS + chunk) * FB_SLOT + lane] = bd;
        part_i[((size_t)f * FB_CHUNKS + chunk) * FB_SLOT + lane] = bi;
      }
    }
    __syncthreads();
  }
}

// one wave per flagged row (the first FB_ROWS): merge the FB_CHUNKS slice lists -- 64 candidates per step, screened against the
// running depth-th best, the few survivors inserted one by one
__global__ __launch_bounds__(256) void exact_fallback_merge_kernel(int depth, const int *__restrict__ flags, const double *__restrict__ part_d,
                                                                   const int *__restrict__ part_i, int64_t *__restrict__ ids,
                                                                   float *__restrict__ dists) {
  const int lane = threadIdx.x & 63;
  const int n_flagged = min(flags[0], FB_ROWS);
  const int total = FB_CHUNKS * depth;
  for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < n_flagged; f += gridDim.x * 4) {
    const int r = flags[1 + f];
    double bd = INFINITY;
    int bi = 0x7fffffff;
    for (int base = 0; base < total; base += 64) {
      const int k = base + lane;
      const int c = k / depth, e = k - c * depth;
      double cv = INFINITY;
      int ci = 0x7fffffff;
      if (k < total) {
        cv = part_d[((size_t)f * FB_CHUNKS + c) * FB_SLOT + e];
        ci = part_i[((size_t)f * FB_CHUNKS + c) * FB_SLOT + e];
      }
      const double tau0 = __shfl(bd, depth - 1, 64);
      const int tau0_i = __shfl(bi, depth - 1, 64);
      unsigned long long mask = __ballot(ci != 0x7fffffff && (cv < tau0 || (cv == tau0 && ci < tau0_i)));
      while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const double xv = __shfl(cv, l, 64);
        const int xi = __shfl(ci, l, 64);
        const double tau = __shfl(bd, depth - 1, 64);
        const int tau_i = __shfl(bi, depth - 1, 64);
        if (!(xv < tau || (xv == tau && xi < tau_i))) continue;
        const bool less = bd < xv || (bd == xv && bi < xi);
        const int pos = __popcll(__ballot(less));
        const double ud = __shfl_up(bd, 1, 64);
        const int ui = __shfl_up(bi, 1, 64);
        if (lane > pos) { bd = ud; bi = ui; }
        if (lane == pos) { bd = xv; bi = xi; }
      }
    }
    if (lane < depth) {
      ids[(size_t)r * depth + lane] = bi == 0x7fffffff ? -1 : (int64_t)bi;
      if (dists) dists[(size_t)r * depth + lane] = (float)bd;
    }
  }
}

// Uncertified owners of the block-minima path (minsel_kernel: more than 64 pool entries under theta, or a block whose fourth
// smallest key is under theta and may hide more): instead of the whole gallery, exactly the entries that CAN be under theta --
// the pool entries <= theta of the safe blocks and EVERY entry of the unsafe ones (an entry that is not among its block's three
// smallest has the block's fourth key below its own).  That set contains the true top-depth; fp64 distances, sorted by
// (distance, index): the same ids as the brute force.  One workgroup per flagged owner, a wave per block at a time.
struct RescanArgs {
  const float *queries, *gallery;
  int ng, d, depth;
  const int *flags;
  const unsigned *keys;
  int R, nblk, bw, nbs, n_src;
  const int *src_base;
  const float *theta;
  int64_t *ids;
  float *dists;
};
__global__ __launch_bounds__(256) void block_rescan_kernel(const RescanArgs PA, const RescanArgs PB, int nblocks_a) {
  const bool second = (int)blockIdx.x >= nblocks_a;
  const RescanArgs &P = second ? PB : PA;
  const int bid = second ? (int)blockIdx.x - nblocks_a : (int)blockIdx.x;
  const int nbl = second ? (int)gridDim.x - nblocks_a : nblocks_a;
  const float *__restrict__ queries = P.queries, *__restrict__ gallery = P.gallery;
  const int ng = P.ng, d = P.d, depth = P.depth, R = P.R, nblk = P.nblk, bw = P.bw, nbs = P.nbs, n_src = P.n_src;
  const int *__restrict__ flags = P.flags;
  const unsigned *__restrict__ keys = P.keys;
  const int *__restrict__ src_base = P.src_base;
  const float *__restrict__ theta = P.theta;
  int64_t *__restrict__ ids = P.ids;
  float *__restrict__ dists = P.dists;
  constexpr int CAP = 4096;               // candidate list of a round: CAP / bw blocks, each at most bw entries
  __shared__ double sd[4][64];
  __shared__ int si[4][64];
  __shared__ int clist[CAP];
  __shared__ int cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_flagged = flags[0];
  const size_t plane = (size_t)nbs * R;
  const int cb = min(CAP / bw, 256);      // blocks per round, a thread per block
  for (int f = bid; f < n_flagged; f += nbl) {     // uniform for the workgroup
    const int r = flags[1 + f];
    const float *q = queries + (size_t)r * d;
    const float th = theta[r];
    double bd = INFINITY;
    int bi = 0x7fffffff;
    auto offer = [&](double cv, int ci) {                   // wave-uniform candidate
      const double tau = __shfl(bd, depth - 1, 64);
      const int tau_i = __shfl(bi, depth - 1, 64);
      if (!(cv < tau || (cv == tau && ci < tau_i))) return;
      const bool less = bd < cv || (bd == cv && bi < ci);
      const int pos = __popcll(__ballot(less));
      const double ud = __shfl_up(bd, 1, 64);
      const int ui = __shfl_up(bi, 1, 64);
      if (lane > pos) { bd = ud; bi = ui; }
      if (lane == pos) { bd = cv; bi = ci; }
    };
    for (int blk0 = 0; blk0 < nblk; blk0 += cb) {
      if (tid == 0) cnt = 0;
      __syncthreads();
      // list the round's candidates: a thread per block, its four keys fetched at once
      const int blk = blk0 + tid;
      if (tid < cb && blk < nblk) {
        const int src = blk / nbs, b_in = blk - src * nbs;
        const size_t o = ((size_t)src * (L2MIN_PLANES - 1) * nbs + blk) * R + r;
        unsigned kk[4];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) kk[pl] = keys[pl * plane + o];
        const int base = (src_base ? src_base[src] : 0) + b_in * bw;
        const int lim = (src_base && src + 1 < n_src) ? src_base[src + 1] : ng;     // the source's rows end here
        if (kk[3] != 0x7F800000u && __uint_as_float(kk[3] & ~127u) <= th) {           // unsafe block: every entry
          const int n_in = min(bw, lim - base);
          if (n_in > 0) {
            const int pos = atomicAdd(&cnt, n_in);
            for (int j = 0; j < n_in; ++j) clist[pos + j] = base + j;
          }
        } else {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            if (kk[pl] != 0x7F800000u && __uint_as_float(kk[pl] & ~127u) <= th) clist[atomicAdd(&cnt, 1)] = base + (int)(kk[pl] & 127u);
        }
      }
      __syncthreads();
      // fp64 distances, four gallery rows per step and wave (their loads in flight together; per row the arithmetic and order
      // of wave_dist64, so a distance has the same bits whichever kernel forms it); the order of the offers does not matter
      const int n_list = cnt;
      for (int c0 = 4 * wave; c0 < n_list; c0 += 16) {
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) jj[u] = clist[min(c0 + u, n_list - 1)];
        double sacc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = lane * 4; c < d; c += 256) {
          const float4 a = *reinterpret_cast<const float4 *>(q + c);
          float4 b[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const float4 *>(gallery + (size_t)jj[u] * d + c);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const double e0 = (double)a.x - (double)b[u].x, e1 = (double)a.y - (double)b[u].y, e2 = (double)a.z - (double)b[u].z,
                         e3 = (double)a.w - (double)b[u].w;
            sacc[u] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) sacc[u] += __shfl_xor(sacc[u], o, 64);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c0 + u < n_list) offer(sacc[u], jj[u]);
      }
      __syncthreads();
    }
    sd[wave][lane] = bd;
    si[wave][lane] = bi;
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < 4; ++w)
        for (int e = 0; e < depth; ++e)
          if (si[w][e] != 0x7fffffff) offer(sd[w][e], si[w][e]);
      if (lane < depth) {
        ids[(size_t)r * depth + lane] = bi == 0x7fffffff ? -1 : (int64_t)bi;
        if (dists) dists[(size_t)r * depth + lane] = (float)bd;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void max_reduce_kernel(const float *__restrict__ x, int n, float *__restrict__ out) {
  __shared__ float part[4];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

// (ids2 != NULL: a second id array of the same shape and targets -- the other direction -- counted into hits2 by the second half of the grid)
__global__ __launch_bounds__(256) void recall_hits_kernel(const int64_t *__restrict__ ids, int n, int depth, int64_t target_offset,
                                                          int k0, int k1, int k2, int k3, int nk, unsigned long long *hits,
                                                          const int64_t *__restrict__ ids2, unsigned long long *hits2) {
  const int half = (int)gridDim.x >> (ids2 ? 1 : 0);
  if (ids2 && (int)blockIdx.x >= half) { ids = ids2; hits = hits2; }
  const int i = ((int)blockIdx.x % half) * 256 + threadIdx.x;
  int rank = 1 << 30;
  if (i < n) {
    const int64_t t = target_offset + i;
    for (int j = 0; j < depth; ++j)
      if (ids[(size_t)i * depth + j] == t) { rank = j; break; }
  }
  const int ks[4] = {k0, k1, k2, k3};
  for (int q = 0; q < nk; ++q) {
    const bool hit = i < n && rank < ks[q];
    const int cnt = __popcll(__ballot(hit));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&hits[q], (unsigned long long)cnt);
  }
}

// terms[i] = lse(sim[i,:]) - sim[i,i];  terms[n + j] = lse(sim[:,j]) - sim[j,j]
__global__ __launch_bounds__(256) void lse_terms_kernel(const float *__restrict__ sim, int n, float *__restrict__ terms) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= 2 * n) return;
  const bool col = w >= n;
  const int i = col ? w - n : w;
  const size_t stride = col ? (size_t)n : 1, base = col ? (size_t)i : (size_t)i * n;
  float mx = -INFINITY;
  for (int c = lane; c < n; c += 64) mx = fmaxf(mx, sim[base + c * stride]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int c = lane; c < n; c += 64) s += expf(sim[base + c * stride] - mx);
  s = wave_sum(s);
  if (lane == 0) terms[w] = mx + logf(s) - sim[(size_t)i * n + i];
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float *__restrict__ terms, int n, float *__restrict__ loss) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < 2 * n; i += 256) s += terms[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = 0.5f * ((part[0] + part[1]) + (part[2] + part[3])) / n;
}

struct SweepWs {
  float *qn, *gn;
  bf16_t *qb, *gb;
  float *dist;
  float *part_d;
  int *part_i;
  int rows_per_block;
  // VTC_SWEEP_EXACT only
  int64_t *cand;
  float *cand_d, *gmax;
  int *flags;
  int cdepth;
  // both directions from one matrix (vtc_l2_topk_bidir): partial column lists, and the EXACT mode's second candidate set
  float *cpart_d;
  int *cpart_i;
  int c_seg, c_total;            // row segments per block = partial lists per column (carried across blocks)
  int64_t *cand2;
  float *cand2_d, *qmax;
  size_t total;
};
constexpr int MAX_SEG = 16;
constexpr int MAX_CSEG = 8;

// candidate list depth of the EXACT mode: 21 spare ranks behind the requested ones, at least 32, at most 64 / n_gallery
int exact_cdepth(int depth, int ng) { return std::min(std::min(64, ng), std::max(32, depth + 21)); }

SweepWs plan(char *ws, int ng, int nq, int d, int precision, int rows_per_block, bool bidir = false) {
  SweepWs s;
  const bool exact = precision == VTC_SWEEP_EXACT;
  if (exact) precision = VTC_SWEEP_BF16X3;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return ws ? ws + o : (char *)nullptr; };
  s.qn = (float *)take((size_t)nq * 4);
  s.gn = (float *)take((size_t)ng * 4);
  const int parts = precision == VTC_SWEEP_BF16X3 ? 3 : (precision == VTC_SWEEP_BF16 ? 1 : 0);
  s.qb = (bf16_t *)take((size_t)nq * d * parts * 2);
  s.gb = (bf16_t *)take((size_t)ng * d * parts * 2);
  int rpb = rows_per_block;
  if (rpb <= 0) {
    // Big blocks: a block's top-k pass costs one list warm-up per (row, segment) wave, and the GEMM a fill/drain
    // per launch, so FEWER, LARGER blocks win over Infinity-Cache residency (measured at 50k x 50k, F32 /
    // BF16X3, one direction: 128 MiB blocks 27.5 / 19.6 ms, 2 GiB blocks 21.4 / 10.4 ms; top-k 1.5 -> 5.0 TB/s).
    const size_t budget = (size_t)2 << 30;
    rpb = (int)std::min<size_t>(budget / ((size_t)ng * 4), (size_t)1 << 20);
    rpb = rpb / 256 * 256;
    if (rpb < 256) rpb = 256;
  }
  if (rpb > nq) rpb = nq;
  s.rows_per_block = rpb;
  s.dist = (float *)take((size_t)rpb * ng * 4);
  s.part_d = (float *)take((size_t)rpb * MAX_SEG * 64 * 4);
  s.part_i = (int *)take((size_t)rpb * MAX_SEG * 64 * 4);
  s.cand = nullptr; s.cand_d = nullptr; s.gmax = nullptr; s.flags = nullptr; s.cdepth = 0;
  if (exact) {
    s.cdepth = 64;                                          // sized for the deepest list (depth is not known to the size query)
    s.cand = (int64_t *)take((size_t)nq * 64 * 8);
    s.cand_d = (float *)take((size_t)nq * 64 * 4);
    s.flags = (int *)take((size_t)(std::max(nq, bidir ? ng : 0) + 1) * 4);   // [0] = number of flagged rows, then their indices
    s.gmax = (float *)take(256);
  }
  s.cpart_d = nullptr; s.cpart_i = nullptr; s.cand2 = nullptr; s.cand2_d = nullptr; s.qmax = nullptr; s.c_seg = 0; s.c_total = 0;
  if (bidir) {
    // enough (strip, segment) workgroups to fill the chip: ~4 per CU
    const int strips = cdiv(ng, 64);
    // (strip, segment) workgroups, four resident per CU: the fewest segments that put ~3 workgroups on every CU --
    // the pass is bound by instruction issue, and every extra segment costs a list initialisation plus its own early
    // insertions (measured at 50k x 50k: 1 / 2 / 3 / 5 segments per block = 4.3 / 4.6 / 5.1 / 6.2 ms per direction)
    s.c_seg = std::max(1, std::min(MAX_CSEG, cdiv(3 * vtcgemm::num_cus(), strips)));
    s.c_seg = std::min(s.c_seg, std::max(1, rpb / 128));                    // at least two tiles per segment
    s.c_total = s.c_seg;                                     // a (column, segment) list is carried from block to block
    s.cpart_d = (float *)take((size_t)ng * s.c_total * 64 * 4);
    s.cpart_i = (int *)take((size_t)ng * s.c_total * 64 * 4);
    if (exact) {
      s.cand2 = (int64_t *)take((size_t)ng * 64 * 8);
      s.cand2_d = (float *)take((size_t)ng * 64 * 4);
      s.qmax = (float *)take(256);
    }
  }
  s.total = off;
  return s;
}

}  // namespace

struct FallbackWs {       // scratch of the chunked fp64 brute force (exact_fallback_chunk_kernel)
  double *part_d;
  int *part_i;
};
struct Rescan {           // the block-minima planes an uncertified owner is re-read from (block_rescan_kernel)
  const unsigned *keys;
  int R, nblk, bw, nbs, n_src;
  const int *src_base;
  const float *theta;
};
static int exact_finish(const float *gallery, const float *queries, int ng, int nq, int d, int depth, int cdepth, const int64_t *cand,
                        const float *cand_d, const float *qn, const float *gn, float *gmax, int *flags, int64_t *ids, float *dists,
                        hipStream_t stream, const int *cand_n, const FallbackWs *fb, const Rescan *rs = nullptr);

// ---- VTC_SWEEP_EXACT, block-minima path ("v2") -----------------------------------------------------------------------
// ONE plain-bf16 distance GEMM whose epilogue keeps, per (query, block of 64 gallery rows), the three smallest distances and
// the fourth as a bound (gemm.hip, EPI_L2MIN) -- and, for vtc_l2_topk_bidir, the same per (gallery row, block of RB queries)
// from the same accumulators; no N x N matrix is written or read.  minsel_kernel turns the block minima into a candidate set
// that provably (given the per-entry error bound exact2_kappa, derived there) contains the query's true top-k (or says that it cannot), exact_rerank_kernel orders it by fp64 distances,
// the uncertified queries are recomputed by fp64 brute force.
namespace {
int exact2_variant() {        // 0: 256 x 256 phased tiles (row blocks of 128); 1: 128 x 128 tiles, two workgroups per CU (row blocks of 64)
  static const int v = [] { const char *e = getenv("VTC_SWEEP_MIN_TILE"); return e ? atoi(e) : 0; }();
  return v;
}
bool exact2_enabled(int ng, int nq, int depth) {
  static const bool off = [] { const char *e = getenv("VTC_SWEEP_EXACT_V1"); return e && e[0] == '1'; }();
  return !off && depth <= 32 && ng >= 1024 && nq >= 1;
}
constexpr int CD2 = 64;       // capacity of a candidate list of the block-minima path
// worst-case |approx - exact| of a key's distance, relative to |q|^2 + max|g|^2.  bf16 keeps 8 significant bits, so its unit
// roundoff under round-to-nearest-even is u = 2^-8 (NOT 2^-9: rounds 1-2 had half this constant -- random embeddings sit far
// inside either bound, coordinated roundings on bf16 midpoints do not: tests/test_gpu_sweep.py adversarial case):
//   q~.g~ - q.g = sum q e_g + g e_q + e_q e_g,  |e| <= u |x|   =>   |.| <= (2u + u^2) |q||g| <= (2u + u^2)(|q|^2 + |g|^2) / 2,
// twice that on the distance: 2^-7 (1 + 2^-9); fp32 accumulation of d products and the two fp32 norms (2 d 2^-24); the 7 index
// bits of the key (2^-16 of a distance <= 2 (|q|^2 + |g|^2)); the epilogue's roundings (slack)
// (round 4: the operand-rounding term is no longer this worst case but the rows' measured |e| -- sweep_prep_kernel / minsel_kernel;
// what remains here are the fp32 terms, relative to |q|^2 + max|g|^2)
float exact2_kappa(int d) { return 2.0f * d / 16777216.0f + 1.0f / 32768.0f + 1e-6f; }

struct Sweep2Ws {
  float *qn, *gn, *gmax, *qmax;      // gmax / qmax: {max |x|^2, max |x~|, max |e|} of the gallery / query side (adjacent 256-byte slots)
  float2 *qst, *gst;                 // (|x~|, |e|) per row
  float *pmax;                       // sweep_prep_kernel's per-block maxima
  bf16_t *qb, *gb;
  unsigned *rowk, *colk;
  int nblk_c, nblk_r, rb;
  int64_t *cand, *cand2;
  int *cand_n, *cand2_n;
  float *theta, *theta2;
  int *flags, *flags2;               // uncertified rows of the row / column direction (count + list)
  FallbackWs fb;
  size_t total;
};
Sweep2Ws plan2(char *ws, int ng, int nq, int d, bool bidir) {
  Sweep2Ws s;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return ws ? ws + o : (char *)nullptr; };
  s.rb = exact2_variant() == 0 ? 128 : 64;
  s.nblk_c = cdiv(ng, 64);
  s.nblk_r = cdiv(nq, s.rb);
  s.qn = (float *)take((size_t)nq * 4);
  s.gn = (float *)take((size_t)ng * 4);
  s.gmax = (float *)take(256);
  s.qmax = (float *)take(256);
  s.qst = (float2 *)take((size_t)nq * 8);
  s.gst = (float2 *)take((size_t)ng * 8);
  s.pmax = (float *)take((size_t)((nq + ng) / 16 + 2) * 8 * 4);
  s.qb = (bf16_t *)take((size_t)nq * d * 2);
  s.gb = (bf16_t *)take((size_t)ng * d * 2);
  s.rowk = (unsigned *)take((size_t)L2MIN_PLANES * s.nblk_c * nq * 4);
  s.colk = bidir ? (unsigned *)take((size_t)L2MIN_PLANES * s.nblk_r * ng * 4) : nullptr;
  s.cand = (int64_t *)take((size_t)nq * CD2 * 8);
  s.cand_n = (int *)take((size_t)nq * 4);
  s.theta = (float *)take((size_t)nq * 4);
  s.cand2 = nullptr; s.cand2_n = nullptr; s.theta2 = nullptr;
  if (bidir) {
    s.cand2 = (int64_t *)take((size_t)ng * CD2 * 8);
    s.cand2_n = (int *)take((size_t)ng * 4);
    s.theta2 = (float *)take((size_t)ng * 4);
  }
  s.flags = (int *)take((size_t)(std::max(nq, bidir ? ng : 0) + 1) * 4);
  s.flags2 = bidir ? (int *)take((size_t)(ng + 1) * 4) : nullptr;
  s.fb.part_d = (double *)take((size_t)FB_ROWS * FB_CHUNKS * FB_SLOT * 8);
  s.fb.part_i = (int *)take((size_t)FB_ROWS * FB_CHUNKS * FB_SLOT * 4);
  s.total = off;
  return s;
}

// ---- recall-only finish of the block-minima sweep (round 5) ----------------------------------------------------------------
// The reference's RecallAtK.compute (model/metric.py:137-161) searches the max(k)+1 nearest gallery rows of every query and then asks
// ONE thing of the sorted list: is the query's own row index among the first k.  That is the RANK of one gallery row,
//     rank_i = #{ j : (|q_i - g_j|^2, j) < (|q_i - g_t|^2, t) },  t = the query's target,        hit at k  <=>  rank_i < k
// (ties by lowest index, as everywhere in this file), and it does not need the sorted top list.  From the key planes of the distance
// GEMM: with d_t the target's exact fp64 distance and eps the row's bound on |key distance - exact| (minsel_kernel's formula),
//   key + eps < d_t   the entry is closer for certain         key - eps > d_t   farther for certain         else: fp64 decides;
// a block whose FOURTH-smallest key is <= d_t + eps may hide unlisted entries in reach of d_t: all its entries go to fp64.
// If the certain ones alone number >= max(k) the query misses at every k and nothing else is looked at -- the fate of most queries of
// an untrained model -- and for a query whose target leads its list (a trained model) almost nothing is in reach: the fp64 re-rank of
// ~20 candidate rows per query (the sweep's largest consumer of bytes) shrinks to the target row and a handful.  Same hits as
// vtc_l2_topk_bidir + vtc_recall_hits_pair, bit for bit in the counters.
// What the recall-only sweep asks of the distance GEMM (gemm.hip, l2min_epilogue).  max(k) <= 16 and many blocks per row (>= 16 384 rows): the
// block's smallest key + the second as a bound in both directions (EPI_L2MIN2: 6 vector instructions per value, half the plane bytes) -- a block
// with two entries in reach of a target goes to fp64 whole, which for a target with r closer entries happens about r^2 / (2 x blocks) of the time:
// rare at 50k (782 / 391 blocks).  At 10k (157 / 79 blocks) it is not, and the COLUMN direction -- blocks of 128 rows: twice as likely and twice
// as dear as the rows' blocks of 64 -- made the finish kernel 93 us of a 0.3 ms sweep on low-recall data: there the columns keep two keys + bound
// (EPI_L2MIN3: rows two planes, columns three; 7 per value).  max(k) > 16: three keys + bound in both.  VTC_SWEEP_PLANES=2|3|4 forces a form.
static int recall_mode(int kmax, int n_total) {
  static const int force = [] { const char *e = getenv("VTC_SWEEP_PLANES"); return e ? atoi(e) : 0; }();
  if (force == 2) return EPI_L2MIN2;
  if (force == 3) return EPI_L2MIN3;
  if (force == 4) return EPI_L2MIN;
  if (kmax > 16) return EPI_L2MIN;
  return n_total < 16384 ? EPI_L2MIN3 : EPI_L2MIN2;
}
struct RankArgs {
  const unsigned *keys;        // [nsrc][NPL][nblk][R], NPL = 4 (EPI_L2MIN: three keys + bound) or 2 (EPI_L2MIN2: one key + bound) -- the kernel's template argument
  int R, nblk, bw;             // owners (queries of this direction), blocks PER SOURCE, entries per block (64 rows / RB columns)
  int nsrc;                    // sources of key planes: 1, or (sharded sweep, column direction) the ranks that each ran a [rows of theirs, R] GEMM
  const int *bounds;           // [nsrc + 1] (device): source s covers the other side's rows [bounds[s], bounds[s + 1]); NULL: one source, [0, ng)
  int tgt_off;                 // owner r is paired with row r + tgt_off of the other side (sharded sweep: the rank's first row; else 0)
  const float *own, *other;    // fp32 rows: [R, d] owners, [ng, d] the other side
  int ng, d;
  const float *own_norm;
  const float2 *own_st;
  const float *other_max;
  float kappa;
  int kmax, nk, k[4];
  unsigned long long *hits;    // [nk], added to
  int *flags;                  // flags[0] = count, flags[1..] = rows left to recall_rank_finish_kernel: r (lists overflowed: brute force) or r | RK_HARD
  int *work;                   // [R][RK_WORK] ints: a deferred row's (closer_safe, n_amb, n_ub, unsafe blocks' first entry[RK_UB] and end[RK_UB], ambiguous entries[RK_AMB])
  int *part;                   // [workgroups of recall_rank_kernel][4]: their hit counts, summed by recall_rank_finish_kernel (one atomic per k
                               // instead of one per wave: 7 500 same-address atomics were 75 of the kernel's 125 us at 10k)
  int nparts;
};
constexpr int RK_OW = 32, RK_CH = 64, RK_AMB = 64, RK_UB = 8;      // owners per workgroup, blocks per chunk, list capacities per owner
constexpr int RK_WORK = 128, RK_HARD = 1 << 30, RK_INLINE = 16;      // ints per deferred row; flag bit; fp64 evaluations a wave does inline
// fp64 |q - g|^2 for up to eight gallery rows at once (the arithmetic and order of wave_dist64: the same doubles whichever kernel forms them)
__device__ __forceinline__ void wave_dist64_x8(const float *__restrict__ q, const float *__restrict__ gallery, const int (&jj)[8], int d, int lane,
                                               double (&out)[8]) {
  double sacc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int c = lane * 4; c < d; c += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(q + c);
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4 *>(gallery + (size_t)(jj[u] < 0 ? 0 : jj[u]) * d + c);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double e0 = (double)a.x - (double)b[u].x, e1 = (double)a.y - (double)b[u].y, e2 = (double)a.z - (double)b[u].z,
                   e3 = (double)a.w - (double)b[u].w;
      sacc[u] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int u = 0; u < 8; ++u) sacc[u] += __shfl_xor(sacc[u], o, 64);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) out[u] = sacc[u];
}
constexpr int RK_TLS = RK_CH * (RK_OW + 1);           // words of one plane's tile in LDS
// A row whose target distance is non-finite adds this to its workgroup's FIRST partial hit count (a workgroup owns RK_OW = 32 rows: the
// real count stays below it); the finish kernel strips it and raises bit 40 of the direction's first counter instead (VTC_RECALL_NONFINITE,
// include/vtc_hip.h) -- the caller learns of NaN / inf embeddings from the counters it reads anyway, without another launch.
constexpr int RK_NONFINITE = 1 << 20;
template <int MAXP>      // planes of the wider of the kernel's two directions: 27 KiB of LDS with two planes (five workgroups per CU), 44 with four (three)
struct RankShared {
  unsigned tl[MAXP * RK_TLS];
  int amb[RK_OW][RK_AMB];
  int ublk[RK_OW][RK_UB], uend[RK_OW][RK_UB];       // unsafe blocks: first entry and end (the block's or its source's)
  int hsum[4][4];
};
// one direction's workgroup; NPL = planes of ITS key layout (the two directions of one sweep may differ: gemm.hip, EPI_L2MIN3)
template <int NPL, typename Shared>
__device__ __forceinline__ void recall_rank_body(const RankArgs &P, int bid, Shared &sh) {
  const int R = P.R, nblk = P.nblk, bw = P.bw, ng = P.ng, d = P.d, kmax = P.kmax;
  unsigned *tl = sh.tl;
  auto &amb = sh.amb;
  auto &ublk = sh.ublk;
  auto &uend = sh.uend;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int toff = P.tgt_off;
  constexpr int OPW = RK_OW / 4;                     // owners per wave
  const int r0 = bid * RK_OW;
  const size_t plane = (size_t)nblk * R;
  // ---- the targets' exact distances and the rows' thresholds (wave-uniform per owner) ----
  double dt[OPW];
  float lo[OPW], hi[OPW];
  {
    // the wave's eight (owner, target) pairs at once: all sixteen row pieces of a K slice are requested before the first is used and the
    // eight xor butterflies interleave (one pair at a time this was eight dependent gather + butterfly latencies: 100 of the kernel's
    // 123 us at 10k).  Per pair the arithmetic and order of wave_dist64 (target of owner r: row r + tgt_off of the other side).
#pragma unroll
    for (int cc = 0; cc < OPW; ++cc) dt[cc] = 0.0;
    for (int c = lane * 4; c < d; c += 256) {
      float4 qa[OPW], gb[OPW];
#pragma unroll
      for (int cc = 0; cc < OPW; ++cc) {
        const int r = min(r0 + OPW * w + cc, R - 1);
        qa[cc] = *reinterpret_cast<const float4 *>(P.own + (size_t)r * d + c);
        gb[cc] = *reinterpret_cast<const float4 *>(P.other + (size_t)min(r + toff, ng - 1) * d + c);
      }
#pragma unroll
      for (int cc = 0; cc < OPW; ++cc) {
        const double e0 = (double)qa[cc].x - (double)gb[cc].x, e1 = (double)qa[cc].y - (double)gb[cc].y, e2 = (double)qa[cc].z - (double)gb[cc].z,
                     e3 = (double)qa[cc].w - (double)gb[cc].w;
        dt[cc] += e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int cc = 0; cc < OPW; ++cc) dt[cc] += __shfl_xor(dt[cc], o, 64);
    }
#pragma unroll
    for (int cc = 0; cc < OPW; ++cc) {
      const int r = min(r0 + OPW * w + cc, R - 1);
      const float2 st = P.own_st[r];
      const float eps = 1.001f * (2.0f * (st.x * P.other_max[2] + st.y * P.other_max[1] + st.y * P.other_max[2]) + P.kappa * (P.own_norm[r] + P.other_max[0]));
      // (two / three planes: the keys are HALF distances -- gemm.hip, EPI_L2MIN2 -- and so are the thresholds; halving is exact)
      constexpr double KS = NPL != 4 ? 0.5 : 1.0;
      lo[cc] = __double2float_rd(KS * (dt[cc] - (double)eps));      // key <  lo  =>  key + eps < d_t
      hi[cc] = __double2float_ru(KS * (dt[cc] + (double)eps));      // key >  hi  =>  key - eps > d_t
    }
  }
  // per LANE (= block of the chunk) counters, reduced over the wave once behind the scan: a ballot + population count per (owner, plane,
  // chunk) for each of them made the scan vector-ALU-bound (17 us per chunk of 64 blocks; 0.9 ms of the 50k sweep)
  int ca_l[OPW], cs_l[OPW], n_amb[OPW], n_ub[OPW];
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) ca_l[cc] = cs_l[cc] = n_amb[cc] = n_ub[cc] = 0;
  const int so = t % RK_OW, sb = t / RK_OW;          // scalar tile loads: this thread's owner and block slice (8 slices)
  const bool vec = (R & 3) == 0;                      // 16-byte loads: four consecutive owners per thread, 8 threads per (plane, block) row
  const int vo = 4 * (t & 7), vb = t >> 3;            // ... this thread's first owner and (block, plane) slice: 32 slices
  for (int src = 0; src < P.nsrc; ++src) {
  const int sbeg = P.bounds ? P.bounds[src] : 0, send = P.bounds ? P.bounds[src + 1] : ng;
  const unsigned *__restrict__ keys = P.keys + (size_t)src * NPL * plane;
  for (int c0 = 0; c0 < nblk; c0 += RK_CH) {
    if (vec) {
      uint4 v[RK_CH * NPL / 32];
#pragma unroll
      for (int q = 0; q < RK_CH * NPL / 32; ++q) {     // (block, plane) pairs vb + 32 q of the chunk: plane = pair % NPL, block = pair / NPL
        const int pr = vb + 32 * q, pl = pr % NPL, bl = pr / NPL, blk = c0 + bl;
        const bool ok = blk < nblk && r0 + vo < R;
        v[q] = ok ? *reinterpret_cast<const uint4 *>(keys + pl * plane + (size_t)blk * R + r0 + vo) : make_uint4(0x7F800000u, 0x7F800000u, 0x7F800000u, 0x7F800000u);
      }
#pragma unroll
      for (int q = 0; q < RK_CH * NPL / 32; ++q) {
        const int pr = vb + 32 * q, pl = pr % NPL, bl = pr / NPL;
        unsigned *dst = &tl[(pl) * RK_TLS + bl * (RK_OW + 1) + vo];
        dst[0] = v[q].x; dst[1] = v[q].y; dst[2] = v[q].z; dst[3] = v[q].w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < RK_CH / 8; ++q) {
        const int bl = sb + 8 * q, blk = c0 + bl;
        const bool ok = blk < nblk && r0 + so < R;
        const size_t o = ok ? (size_t)blk * R + r0 + so : 0;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) tl[(pl) * RK_TLS + bl * (RK_OW + 1) + so] = ok ? keys[pl * plane + o] : 0x7F800000u;
      }
    }
    __syncthreads();
    const int blk = c0 + lane;
    const int base = sbeg + blk * bw;
#pragma unroll
    for (int cc = 0; cc < OPW; ++cc) {
      const int o = OPW * w + cc, tg = r0 + o + toff;
      const unsigned k3 = tl[(NPL - 1) * RK_TLS + lane * (RK_OW + 1) + o];            // the block's bound: its NPL-th smallest key
      const bool unsafe = k3 != 0x7F800000u && __uint_as_float(k3 & ~127u) <= hi[cc];
      // the bound is an entry too (the block's NPL-th smallest): closer for certain, it raises the LOWER bound on the rank that ends a query as a
      // miss without any fp64 (with two planes a query whose few closer entries share blocks would otherwise go to fp64 for a rank >= max k)
      ca_l[cc] += (unsafe && __uint_as_float(k3 & ~127u) < lo[cc] && base + (int)(k3 & 127u) != tg) ? 1 : 0;
      unsigned kk[NPL - 1];
      bool am[NPL - 1];
      bool any_am = false;
#pragma unroll
      for (int pl = 0; pl < NPL - 1; ++pl) {
        kk[pl] = tl[(pl) * RK_TLS + lane * (RK_OW + 1) + o];
        const float v = __uint_as_float(kk[pl] & ~127u);
        const bool valid = kk[pl] != 0x7F800000u && base + (int)(kk[pl] & 127u) != tg;     // (the target itself is not counted)
        const bool dc = valid && v < lo[cc];
        am[pl] = valid && !dc && !(v > hi[cc]) && !unsafe;                                  // (an unsafe block's entries all go to fp64 below)
        any_am |= am[pl];
        ca_l[cc] += dc ? 1 : 0;
        cs_l[cc] += (dc && !unsafe) ? 1 : 0;
      }
      const unsigned long long um = __ballot(unsafe);
      if (um) {                                                           // wave-uniform, rare
        const int pos = n_ub[cc] + __popcll(um & ((1ull << lane) - 1ull));
        if (unsafe && pos < RK_UB) { ublk[o][pos] = base; uend[o][pos] = min(base + bw, send); }
        n_ub[cc] += __popcll(um);
      }
      if (__ballot(any_am)) {                                             // wave-uniform, rare: entries within eps of the target's distance
#pragma unroll
        for (int pl = 0; pl < NPL - 1; ++pl) {
          const unsigned long long mask = __ballot(am[pl]);
          const int pos = n_amb[cc] + __popcll(mask & ((1ull << lane) - 1ull));
          if (am[pl] && pos < RK_AMB) amb[o][pos] = base + (int)(kk[pl] & 127u);
          n_amb[cc] += __popcll(mask);
        }
      }
    }
    __syncthreads();
  }
  }
  int closer_all[OPW], closer_safe[OPW];
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) {
    int x = ca_l[cc], y = cs_l[cc];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
    closer_all[cc] = x; closer_safe[cc] = y;
  }
  // ---- settle: certain misses, fp64 for what is in reach, the rest to the fallback ----
  int hit_cnt[4] = {0, 0, 0, 0};
#pragma unroll
  for (int cc = 0; cc < OPW; ++cc) {
    const int o = OPW * w + cc, r = r0 + o, tg = r + toff;
    if (r >= R) continue;                                                 // wave-uniform
    if (closer_all[cc] >= kmax) continue;                                 // a miss at every k
    // a non-finite target distance (NaN / inf in the query's or its target's embedding): every comparison below would be false and the
    // rank would read 0 -- a hit at every k.  An exact search never returns such a target (no distance compares below NaN): a miss.
    if (!(dt[cc] < (double)INFINITY)) { hit_cnt[0] += RK_NONFINITE; continue; }       // (... and said: recall_rank_finish_kernel raises VTC_RECALL_NONFINITE in the counters)
    if (n_amb[cc] > RK_AMB || n_ub[cc] > RK_UB) {
      if (lane == 0) P.flags[1 + atomicAdd(P.flags, 1)] = r;
      continue;
    }
    if (n_amb[cc] + n_ub[cc] * bw > RK_INLINE) {
      // a row with much in reach of its target (the target sits inside the bulk of its row's distances, yet fewer than max(k) entries are
      // closer for certain): its fp64 work -- every entry of its unsafe blocks -- is a long serial chain for ONE wave that has seven more
      // owners to do (the stragglers were the whole kernel time: 127 us at 10k for 0.3 % of the rows).  Deferred with its lists:
      // recall_rank_finish_kernel gives it a workgroup of its own.
      int *wk = P.work + (size_t)r * RK_WORK;
      if (lane == 0) { wk[0] = closer_safe[cc]; wk[1] = n_amb[cc]; wk[2] = n_ub[cc]; }
      if (lane < n_ub[cc]) { wk[3 + lane] = ublk[o][lane]; wk[3 + RK_UB + lane] = uend[o][lane]; }
      if (lane < n_amb[cc]) wk[3 + 2 * RK_UB + lane] = amb[o][lane];
      if (lane == 0) P.flags[1 + atomicAdd(P.flags, 1)] = r | RK_HARD;
      continue;
    }
    const float *q = P.own + (size_t)r * d;
    int rank = closer_safe[cc];
    auto count_group = [&](const int (&jj)[8]) {
      double dd[8];
      wave_dist64_x8(q, P.other, jj, d, lane, dd);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (jj[u] >= 0 && (dd[u] < dt[cc] || (dd[u] == dt[cc] && jj[u] < tg))) ++rank;
    };
    for (int c0 = 0; c0 < n_amb[cc] && rank < kmax; c0 += 8) {
      int jj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) jj[u] = c0 + u < n_amb[cc] ? amb[o][c0 + u] : -1;
      count_group(jj);
    }
    for (int ub = 0; ub < n_ub[cc] && rank < kmax; ++ub) {
      const int base = ublk[o][ub], end = uend[o][ub];
      for (int c0 = 0; c0 < bw && rank < kmax; c0 += 8) {
        int jj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = base + c0 + u;
          jj[u] = (j < end && j != tg) ? j : -1;
        }
        count_group(jj);
      }
    }
#pragma unroll
    for (int qk = 0; qk < 4; ++qk)
      if (qk < P.nk && rank < P.k[qk]) ++hit_cnt[qk];
  }
  auto &hsum = sh.hsum;
  if (lane == 0) {
#pragma unroll
    for (int qk = 0; qk < 4; ++qk) hsum[w][qk] = hit_cnt[qk];
  }
  __syncthreads();
  if (t < 4) P.part[(size_t)bid * 4 + t] = hsum[0][t] + hsum[1][t] + hsum[2][t] + hsum[3][t];
}
template <int NPLA, int NPLB>
__global__ __launch_bounds__(256) void recall_rank_kernel(const RankArgs PA, const RankArgs PB, int nblocks_a) {
  __shared__ RankShared<(NPLA > NPLB ? NPLA : NPLB)> sh;
  if ((int)blockIdx.x < nblocks_a) recall_rank_body<NPLA>(PA, (int)blockIdx.x, sh);
  else recall_rank_body<NPLB>(PB, (int)blockIdx.x - nblocks_a, sh);
}
// The rows recall_rank_kernel left over, one workgroup per row: a DEFERRED row (r | RK_HARD) comes with its lists -- the waves split its
// fp64 evaluations; a row whose lists overflowed (dense near-ties around the target) gets its rank by fp64 brute force over the other side.
constexpr int RK_FW = 8;        // waves of a finish workgroup: a deferred row is 64 - 128 fp64 distances of 2 KB rows, latency-bound -- eight at once per wave
__global__ __launch_bounds__(64 * RK_FW) void recall_rank_finish_kernel(const RankArgs PA, const RankArgs PB, int nblocks_a) {
  const bool second = (int)blockIdx.x >= nblocks_a;
  const RankArgs &P = second ? PB : PA;
  const int bid = second ? (int)blockIdx.x - nblocks_a : (int)blockIdx.x;
  const int nbl = second ? (int)gridDim.x - nblocks_a : nblocks_a;
  __shared__ int part[RK_FW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (bid == 0) {            // this direction's first workgroup: the rank kernel's per-workgroup hit counts -> one atomic per k
    int acc[4] = {0, 0, 0, 0};
    int nf = 0;
    for (int i = threadIdx.x; i < P.nparts; i += 64 * RK_FW) {
      const int4 v = *reinterpret_cast<const int4 *>(P.part + (size_t)i * 4);
      nf |= v.x >> 20;                                   // rows with a non-finite target distance (RK_NONFINITE each)
      acc[0] += v.x & (RK_NONFINITE - 1); acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
    }
    if (__ballot(nf != 0) != 0 && lane == 0) atomicOr(&P.hits[0], 1ull << 40);       // VTC_RECALL_NONFINITE
    __shared__ int red[RK_FW][4];
#pragma unroll
    for (int qk = 0; qk < 4; ++qk) {
      int x = acc[qk];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
      if (lane == 0) red[w][qk] = x;
    }
    __syncthreads();
    if ((int)threadIdx.x < P.nk) {
      int x = 0;
#pragma unroll
      for (int ww = 0; ww < RK_FW; ++ww) x += red[ww][threadIdx.x];
      if (x) atomicAdd(&P.hits[threadIdx.x], (unsigned long long)x);
    }
    __syncthreads();
  }
  const int n_flagged = P.flags[0];
  for (int f = bid; f < n_flagged; f += nbl) {          // uniform for the workgroup; normally few trips
    const int fr = P.flags[1 + f];
    const bool hard = (fr & RK_HARD) != 0;
    const int r = fr & ~RK_HARD, tg = r + P.tgt_off;
    const float *q = P.own + (size_t)r * P.d;
    const double dt = wave_dist64(q, P.other + (size_t)tg * P.d, P.d, lane);
    const bool dt_ok = dt < (double)INFINITY;      // (the rank kernel never flags a row with a non-finite target distance; kept for the brute-force form)
    int cnt = 0, base_rank = 0;
    auto count_group = [&](const int (&jj)[8]) {
      double dd[8];
      wave_dist64_x8(q, P.other, jj, P.d, lane, dd);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (jj[u] >= 0 && (dd[u] < dt || (dd[u] == dt && jj[u] < tg))) ++cnt;
    };
    if (hard) {
      const int *wk = P.work + (size_t)r * RK_WORK;
      base_rank = wk[0];
      const int n_amb = wk[1], n_ub = wk[2];
      const int n_eval = n_amb + n_ub * P.bw;           // evaluation e: e < n_amb -> ambiguous entry e; else entry (e - n_amb) % bw of unsafe block (e - n_amb) / bw
      for (int e0 = 8 * w; e0 < n_eval; e0 += 8 * RK_FW) {
        int jj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u;
          int j = -1;
          if (e < n_amb) j = wk[3 + 2 * RK_UB + e];
          else if (e < n_eval) {
            const int x = e - n_amb, ub = x / P.bw;
            j = wk[3 + ub] + (x - ub * P.bw);
            if (j >= wk[3 + RK_UB + ub] || j == tg) j = -1;
          }
          jj[u] = j;
        }
        count_group(jj);
      }
    } else {
      for (int j0 = 8 * w; j0 < P.ng; j0 += 8 * RK_FW) {
        int jj[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) jj[u] = (j0 + u < P.ng && j0 + u != tg) ? j0 + u : -1;
        count_group(jj);
      }
    }
    if (lane == 0) part[w] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      int rank = base_rank;
#pragma unroll
      for (int ww = 0; ww < RK_FW; ++ww) rank += part[ww];
#pragma unroll
      for (int qk = 0; qk < 4; ++qk)          // (constant indices: a run-time index into P.k[] sends the argument block to scratch)
        if (qk < P.nk && dt_ok && rank < P.k[qk]) atomicAdd(&P.hits[qk], 1ull);
    }
    __syncthreads();
  }
}

// gallery a [na], queries b [nb]; ids_a2b != nullptr: also the transposed direction
// colk_out != NULL (sharded sweep, rank-local rows): the column planes go to the caller's [4, nblk_r_pad, na] buffer and the
// column direction is NOT finished here (vtc_l2_sweep_shard_cols does, after the exchange)
int exact2_impl(const float *a, const float *b, int na, int nb, int d, int depth, int64_t *ids_b2a, float *dists_b2a, int64_t *ids_a2b,
                float *dists_a2b, const Sweep2Ws &s, hipStream_t stream, unsigned *colk_out = nullptr, int nblk_r_pad = 0) {
  {   // norms, bf16 operands, rounding-error statistics and their maxima of both sets: one launch (rounds 1-3: six)
    ProfScope prof(VTC_PROF_TOPK, (double)(na + nb) * d * 6, stream);
    const PrepSide A{b, s.qb, s.qn, s.qst, s.qmax, nb}, B{a, s.gb, s.gn, s.gst, s.gmax, na};
    launch_sweep_prep(A, B, d, s.pmax, stream);
  }
  VTC_LAUNCH_CHECK("l2_topk prologue");
  GemmEpi e;
  e.mode = EPI_L2MIN; e.out_dtype = VTC_F32; e.rown = s.qn; e.coln = s.gn;
  e.rowk = s.rowk; e.colk = ids_a2b ? s.colk : nullptr; e.nblk_c = s.nblk_c; e.nblk_r = s.nblk_r; e.rb = s.rb;
  if (colk_out) {
    e.colk = colk_out; e.nblk_r = nblk_r_pad;
    // blocks this rank has no rows for (shards differ by a row): +inf keys
    for (int pl = 0; pl < L2MIN_PLANES && nblk_r_pad > s.nblk_r; ++pl)
      (void)hipMemsetD32Async((hipDeviceptr_t)(colk_out + ((size_t)pl * nblk_r_pad + s.nblk_r) * na), 0x7F800000,
                              (size_t)(nblk_r_pad - s.nblk_r) * na, stream);
  }
  if (int rc = launch_gemm(s.qb, s.gb, nullptr, nullptr, nb, na, d, VTC_BF16, e, stream)) return rc;
  const float kappa = exact2_kappa(d);
  const MinselArgs m1 = minsel_args(s.rowk, nb, s.nblk_c, 64, s.nblk_c, nullptr, depth, s.qn, s.qst, s.gmax, kappa, s.cand, s.cand_n, s.theta);
  const Rescan rs1{s.rowk, nb, s.nblk_c, 64, s.nblk_c, 1, nullptr, s.theta};
  if (!ids_a2b) {
    {
      ProfScope prof(VTC_PROF_TOPK, (double)(L2MIN_PLANES + 1) * s.nblk_c * nb * 4, stream);
      launch_minsel(m1, nullptr, stream);
    }
    VTC_LAUNCH_CHECK("minsel");
    return exact_finish(a, b, na, nb, d, depth, CD2, s.cand, nullptr, s.qn, s.gn, s.gmax, s.flags, ids_b2a, dists_b2a, stream, s.cand_n, &s.fb, &rs1);
  }
  // both directions: ONE launch per stage (rounds 2-3: two) -- the stages of the two directions are independent, and at 10k x 10k
  // a direction alone leaves CUs idle (313 minsel workgroups, a re-rank wave per row)
  const MinselArgs m2 = minsel_args(s.colk, na, s.nblk_r, s.rb, s.nblk_r, nullptr, depth, s.gn, s.gst, s.qmax, kappa, s.cand2, s.cand2_n, s.theta2);
  {
    ProfScope prof(VTC_PROF_TOPK, (double)(L2MIN_PLANES + 1) * ((double)s.nblk_c * nb + (double)s.nblk_r * na) * 4, stream);
    launch_minsel(m1, &m2, stream);
  }
  VTC_LAUNCH_CHECK("minsel");
  (void)hipMemsetAsync(s.flags, 0, (size_t)((char *)s.flags2 - (char *)s.flags) + sizeof(int), stream);     // both counters: one fill (the lists between them are rewritten anyway)
  const RerankArgs r1{b, a, nb, na, d, s.cand, nullptr, CD2, depth, s.qn, s.gmax, 0.f, ids_b2a, dists_b2a, s.flags, s.cand_n};
  const RerankArgs r2{a, b, na, nb, d, s.cand2, nullptr, CD2, depth, s.gn, s.qmax, 0.f, ids_a2b, dists_a2b, s.flags2, s.cand2_n};
  {
    ProfScope prof(VTC_PROF_TOPK, (double)(na + nb) * (2.0 * depth + 1.0) * d * 4, stream);
    hipLaunchKernelGGL(exact_rerank_kernel, dim3(cdiv(nb, 4) + cdiv(na, 4)), dim3(256), 0, stream, r1, r2, cdiv(nb, 4));
  }
  {
    ProfScope prof(VTC_PROF_TOPK, 0.0, stream);
    const RescanArgs q1{b, a, na, d, depth, s.flags, s.rowk, nb, s.nblk_c, 64, s.nblk_c, 1, nullptr, s.theta, ids_b2a, dists_b2a};
    const RescanArgs q2{a, b, nb, d, depth, s.flags2, s.colk, na, s.nblk_r, s.rb, s.nblk_r, 1, nullptr, s.theta2, ids_a2b, dists_a2b};
    const int g1 = std::min(nb, 1024), g2 = std::min(na, 1024);
    hipLaunchKernelGGL(block_rescan_kernel, dim3(g1 + g2), dim3(256), 0, stream, q1, q2, g1);
  }
  VTC_LAUNCH_CHECK("l2_topk exact (both directions)");
  return 0;
}

// R@K hit counters of BOTH directions of n paired rows (a_i <-> b_i) without the sorted lists: prologue, ONE distance GEMM, ONE
// rank launch (+ the fallback launch, normally empty).  hits_b_from_a[j] += #{ i : rank of a_i among the a's for query b_i < k_j }
// (= RecallAtK.compute(a, b)), hits_a_from_b the transposed direction (= compute(b, a)).
int recall_bidir_impl(const float *a, const float *b, int n, int d, const int *k_vals, int nk, unsigned long long *hits_b_from_a,
                      unsigned long long *hits_a_from_b, const Sweep2Ws &s, hipStream_t stream) {
  {
    ProfScope prof(VTC_PROF_TOPK, (double)2 * n * d * 6, stream);
    const PrepSide A{b, s.qb, s.qn, s.qst, s.qmax, n}, B{a, s.gb, s.gn, s.gst, s.gmax, n};
    launch_sweep_prep(A, B, d, s.pmax, stream, s.flags, s.flags2);       // ... and the two fallback counters zeroed
  }
  VTC_LAUNCH_CHECK("l2_recall prologue");
  int kmax = 0;
  for (int i = 0; i < nk; ++i) kmax = std::max(kmax, k_vals[i]);
  const int mode = recall_mode(kmax, n);
  GemmEpi e;
  e.mode = mode; e.out_dtype = VTC_F32; e.rown = s.qn; e.coln = s.gn;
  e.rowk = s.rowk; e.colk = s.colk; e.nblk_c = s.nblk_c; e.nblk_r = s.nblk_r; e.rb = s.rb;
  if (int rc = launch_gemm(s.qb, s.gb, nullptr, nullptr, n, n, d, VTC_BF16, e, stream)) return rc;
  const float kappa = exact2_kappa(d);
  kmax = 0;
  static_assert(RK_WORK * sizeof(int) <= CD2 * sizeof(int64_t) && 3 + 2 * RK_UB + RK_AMB <= RK_WORK, "a deferred row's lists fit its candidate-list slot");
  const int nwg = cdiv(n, RK_OW);            // (4 ints per workgroup in the cand_n arrays: n / 8 <= n)
  RankArgs r1{s.rowk, n, s.nblk_c, 64, 1, nullptr, 0, b, a, n, d, s.qn, s.qst, s.gmax, kappa, 0, nk, {0, 0, 0, 0}, hits_b_from_a, s.flags, (int *)s.cand, s.cand_n, nwg};
  RankArgs r2{s.colk, n, s.nblk_r, s.rb, 1, nullptr, 0, a, b, n, d, s.gn, s.gst, s.qmax, kappa, 0, nk, {0, 0, 0, 0}, hits_a_from_b, s.flags2, (int *)s.cand2, s.cand2_n, nwg};
  for (int i = 0; i < nk; ++i) { r1.k[i] = r2.k[i] = k_vals[i]; kmax = std::max(kmax, k_vals[i]); }
  r1.kmax = r2.kmax = kmax;
  {
    ProfScope prof(VTC_PROF_TOPK, ((double)l2min_row_planes(mode) * s.nblk_c + (double)l2min_col_planes(mode) * s.nblk_r) * n * 4 + 4.0 * n * d * 4, stream);
    const int nb = cdiv(n, RK_OW);
    if (mode == EPI_L2MIN2) hipLaunchKernelGGL((recall_rank_kernel<2, 2>), dim3(2 * nb), dim3(256), 0, stream, r1, r2, nb);
    else if (mode == EPI_L2MIN3) hipLaunchKernelGGL((recall_rank_kernel<2, 3>), dim3(2 * nb), dim3(256), 0, stream, r1, r2, nb);
    else hipLaunchKernelGGL((recall_rank_kernel<4, 4>), dim3(2 * nb), dim3(256), 0, stream, r1, r2, nb);
  }
  {
    ProfScope prof(VTC_PROF_TOPK, 0.0, stream);
    const int g = std::min(n, 1024);
    hipLaunchKernelGGL(recall_rank_finish_kernel, dim3(2 * g), dim3(64 * RK_FW), 0, stream, r1, r2, g);
  }
  VTC_LAUNCH_CHECK("l2_recall_bidir");
  return 0;
}
}  // namespace

// ---- sharded sweep: ONE [N/G, N] distance GEMM per rank for both directions (include/vtc_hip.h) ---------------------------
extern "C" int vtc_l2_sweep_row_block(void) { return exact2_variant() == 0 ? 128 : 64; }

extern "C" int vtc_l2_sweep_shard_supported(int n_total, int n_local, int depth) {
  return exact2_enabled(n_total, n_local, depth) && n_local >= 1 && depth <= n_total;
}

extern "C" size_t vtc_l2_sweep_shard_workspace_bytes(int n_total, int n_local, int d) {
  // rows phase: plan2 of (gallery n_total, queries n_local); cols phase: norms of both sides + candidates of n_local queries
  const size_t rows = plan2(nullptr, n_total, n_local, d, false).total;
  return rows + align_up((size_t)n_total * 4, 256) + 512;
}

extern "C" int vtc_l2_sweep_shard_rows(const float *a_all, const float *b_local, int n_total, int n_local, int d, int depth,
                                       int64_t *ids, float *dists, unsigned *col_planes, int nblk_pad, void *ws, size_t ws_bytes,
                                       void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(a_all && b_local && ids && col_planes && ws, "l2_sweep_shard_rows: null argument");
  VTC_CHECK(d > 0 && d % 64 == 0, "l2_sweep_shard_rows: d=%d must be a positive multiple of 64", d);
  VTC_CHECK(vtc_l2_sweep_shard_supported(n_total, n_local, depth), "l2_sweep_shard_rows: unsupported shape (n_total=%d n_local=%d depth=%d)",
            n_total, n_local, depth);
  Sweep2Ws s2 = plan2((char *)ws, n_total, n_local, d, false);
  VTC_CHECK(nblk_pad >= s2.nblk_r, "l2_sweep_shard_rows: nblk_pad=%d < %d row blocks", nblk_pad, s2.nblk_r);
  VTC_CHECK(ws_bytes >= vtc_l2_sweep_shard_workspace_bytes(n_total, n_local, d), "l2_sweep_shard_rows: workspace too small");
  return exact2_impl(a_all, b_local, n_total, n_local, d, depth, ids, dists, nullptr, nullptr, s2, stream, col_planes, nblk_pad);
}

extern "C" int vtc_l2_sweep_shard_cols(const float *b_all, const float *a_local, int n_total, int n_local, int d, int depth,
                                       const unsigned *planes, int n_src, int nblk_pad, const int *src_base, int64_t *ids,
                                       float *dists, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(b_all && a_local && planes && src_base && ids && ws, "l2_sweep_shard_cols: null argument");
  VTC_CHECK(d > 0 && d % 64 == 0, "l2_sweep_shard_cols: d=%d must be a positive multiple of 64", d);
  VTC_CHECK(vtc_l2_sweep_shard_supported(n_total, n_local, depth) && n_src >= 1 && nblk_pad >= 1,
            "l2_sweep_shard_cols: unsupported shape (n_total=%d n_local=%d depth=%d n_src=%d)", n_total, n_local, depth, n_src);
  VTC_CHECK(ws_bytes >= vtc_l2_sweep_shard_workspace_bytes(n_total, n_local, d), "l2_sweep_shard_cols: workspace too small");
  // gallery = b_all (n_total), queries = a_local (n_local): the plan's qn/gn/cand/flags/fallback areas fit as they are
  Sweep2Ws s = plan2((char *)ws, n_total, n_local, d, false);
  const int rb = vtc_l2_sweep_row_block();
  {   // statistics only (the operands were rounded by the ranks that ran the GEMMs -- with this same rounding)
    ProfScope prof(VTC_PROF_TOPK, (double)(n_total + n_local) * d * 4, stream);
    const PrepSide A{a_local, nullptr, s.qn, s.qst, s.qmax, n_local}, B{b_all, nullptr, s.gn, s.gst, s.gmax, n_total};
    launch_sweep_prep(A, B, d, s.pmax, stream);
  }
  const float kappa = exact2_kappa(d);
  {
    ProfScope prof(VTC_PROF_TOPK, (double)(L2MIN_PLANES + 1) * n_src * nblk_pad * n_local * 4, stream);
    const MinselArgs m = minsel_args(planes, n_local, n_src * nblk_pad, rb, nblk_pad, src_base, depth, s.qn, s.qst, s.gmax, kappa, s.cand, s.cand_n, s.theta);
    launch_minsel(m, nullptr, stream);
  }
  VTC_LAUNCH_CHECK("minsel shard cols");
  const Rescan rs{planes, n_local, n_src * nblk_pad, rb, nblk_pad, n_src, src_base, s.theta};
  return exact_finish(b_all, a_local, n_total, n_local, d, depth, CD2, s.cand, nullptr, s.qn, s.gn, s.gmax, s.flags, ids, dists, stream,
                      s.cand_n, &s.fb, &rs);
}

// ---- the same exchange with the recall-only finish (round 5): hit counters instead of sorted lists ---------------------------
// One direction of recall_rank_kernel + recall_rank_finish_kernel (the second argument block is not used: every workgroup is "first").
static void launch_rank_one(const RankArgs &r, int npl, hipStream_t stream) {
  {
    ProfScope prof(VTC_PROF_TOPK, (double)npl * r.nsrc * r.nblk * r.R * 4 + 2.0 * r.R * r.d * 4, stream);
    const int nb = cdiv(r.R, RK_OW);
    if (npl == 2) hipLaunchKernelGGL((recall_rank_kernel<2, 2>), dim3(nb), dim3(256), 0, stream, r, r, nb);
    else if (npl == 3) hipLaunchKernelGGL((recall_rank_kernel<3, 3>), dim3(nb), dim3(256), 0, stream, r, r, nb);
    else hipLaunchKernelGGL((recall_rank_kernel<4, 4>), dim3(nb), dim3(256), 0, stream, r, r, nb);
  }
  {
    ProfScope prof(VTC_PROF_TOPK, 0.0, stream);
    const int g = std::min(r.R, 1024);
    hipLaunchKernelGGL(recall_rank_finish_kernel, dim3(g), dim3(64 * RK_FW), 0, stream, r, r, g);
  }
}

static int fill_k(RankArgs &r, const int *k_vals, int nk) {
  r.nk = nk; r.kmax = 0;
  for (int i = 0; i < 4; ++i) r.k[i] = 0;
  for (int i = 0; i < nk; ++i) { r.k[i] = k_vals[i]; r.kmax = std::max(r.kmax, k_vals[i]); }
  return r.kmax;
}

extern "C" int vtc_l2_recall_planes(const int *k_vals, int nk, int n_total) {
  int kmax = 0;
  for (int i = 0; k_vals && i < nk; ++i) kmax = std::max(kmax, k_vals[i]);
  return l2min_col_planes(recall_mode(kmax, n_total));       // (the planes that travel: the column direction's)
}

extern "C" int vtc_l2_recall_shard_supported(int n_total, int n_local, int d) {
  return d > 0 && d % 64 == 0 && n_local >= 1 && n_local <= n_total && exact2_enabled(n_total, n_local, 1);
}

extern "C" int vtc_l2_recall_shard_rows(const float *a_all, const float *b_local, int n_total, int n_local, int row_base, int d,
                                        const int *k_vals, int nk, long long *hits_b_from_a, unsigned *col_planes, int nblk_pad, void *ws,
                                        size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(a_all && b_local && k_vals && hits_b_from_a && col_planes && ws, "l2_recall_shard_rows: null argument");
  VTC_CHECK(vtc_l2_recall_shard_supported(n_total, n_local, d), "l2_recall_shard_rows: unsupported shape (n_total=%d n_local=%d d=%d)", n_total, n_local, d);
  VTC_CHECK(row_base >= 0 && row_base + n_local <= n_total, "l2_recall_shard_rows: rows [%d, %d) outside [0, %d)", row_base, row_base + n_local, n_total);
  VTC_CHECK(nk >= 1 && nk <= 4, "l2_recall_shard_rows: nk=%d must be in 1..4", nk);
  for (int i = 0; i < nk; ++i) VTC_CHECK(k_vals[i] >= 1 && k_vals[i] <= n_total, "l2_recall_shard_rows: k=%d must be in 1..%d", k_vals[i], n_total);
  Sweep2Ws s = plan2((char *)ws, n_total, n_local, d, false);
  VTC_CHECK(nblk_pad >= s.nblk_r, "l2_recall_shard_rows: nblk_pad=%d < %d row blocks", nblk_pad, s.nblk_r);
  VTC_CHECK(ws_bytes >= vtc_l2_sweep_shard_workspace_bytes(n_total, n_local, d), "l2_recall_shard_rows: workspace too small");
  {
    ProfScope prof(VTC_PROF_TOPK, (double)(n_total + n_local) * d * 6, stream);
    const PrepSide A{b_local, s.qb, s.qn, s.qst, s.qmax, n_local}, B{a_all, s.gb, s.gn, s.gst, s.gmax, n_total};
    launch_sweep_prep(A, B, d, s.pmax, stream, s.flags, nullptr);
  }
  VTC_LAUNCH_CHECK("l2_recall_shard_rows prologue");
  int kmax = 0;
  for (int i = 0; i < nk; ++i) kmax = std::max(kmax, k_vals[i]);
  const int mode = recall_mode(kmax, n_total), npl = l2min_col_planes(mode);
  GemmEpi e;
  e.mode = mode; e.out_dtype = VTC_F32; e.rown = s.qn; e.coln = s.gn;
  e.rowk = s.rowk; e.colk = col_planes; e.nblk_c = s.nblk_c; e.nblk_r = nblk_pad; e.rb = s.rb;
  for (int pl = 0; pl < npl && nblk_pad > s.nblk_r; ++pl)     // blocks this rank has no rows for (shards differ by a row): +inf keys
    (void)hipMemsetD32Async((hipDeviceptr_t)(col_planes + ((size_t)pl * nblk_pad + s.nblk_r) * n_total), 0x7F800000,
                            (size_t)(nblk_pad - s.nblk_r) * n_total, stream);
  if (int rc = launch_gemm(s.qb, s.gb, nullptr, nullptr, n_local, n_total, d, VTC_BF16, e, stream)) return rc;
  RankArgs r{s.rowk, n_local, s.nblk_c, 64, 1, nullptr, row_base, b_local, a_all, n_total, d, s.qn, s.qst, s.gmax, exact2_kappa(d), 0, nk, {0, 0, 0, 0},
             (unsigned long long *)hits_b_from_a, s.flags, (int *)s.cand, s.cand_n, cdiv(n_local, RK_OW)};
  fill_k(r, k_vals, nk);
  launch_rank_one(r, l2min_row_planes(mode), stream);
  VTC_LAUNCH_CHECK("l2_recall_shard_rows");
  return 0;
}

extern "C" int vtc_l2_recall_shard_cols(const float *b_all, const float *a_local, int n_total, int n_local, int row_base, int d,
                                        const int *k_vals, int nk, const unsigned *planes, int n_src, int nblk_pad, const int *src_bounds,
                                        long long *hits_a_from_b, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(b_all && a_local && k_vals && planes && src_bounds && hits_a_from_b && ws, "l2_recall_shard_cols: null argument");
  VTC_CHECK(vtc_l2_recall_shard_supported(n_total, n_local, d) && n_src >= 1 && nblk_pad >= 1,
            "l2_recall_shard_cols: unsupported shape (n_total=%d n_local=%d d=%d n_src=%d)", n_total, n_local, d, n_src);
  VTC_CHECK(row_base >= 0 && row_base + n_local <= n_total, "l2_recall_shard_cols: rows [%d, %d) outside [0, %d)", row_base, row_base + n_local, n_total);
  VTC_CHECK(nk >= 1 && nk <= 4, "l2_recall_shard_cols: nk=%d must be in 1..4", nk);
  for (int i = 0; i < nk; ++i) VTC_CHECK(k_vals[i] >= 1 && k_vals[i] <= n_total, "l2_recall_shard_cols: k=%d must be in 1..%d", k_vals[i], n_total);
  VTC_CHECK(ws_bytes >= vtc_l2_sweep_shard_workspace_bytes(n_total, n_local, d), "l2_recall_shard_cols: workspace too small");
  // owners = a_local (this rank's columns of every source's GEMM), the other side = b_all: the plan's areas fit as they are
  Sweep2Ws s = plan2((char *)ws, n_total, n_local, d, false);
  {   // statistics only (the operands were rounded by the ranks that ran the GEMMs -- with this same rounding)
    ProfScope prof(VTC_PROF_TOPK, (double)(n_total + n_local) * d * 4, stream);
    const PrepSide A{a_local, nullptr, s.qn, s.qst, s.qmax, n_local}, B{b_all, nullptr, s.gn, s.gst, s.gmax, n_total};
    launch_sweep_prep(A, B, d, s.pmax, stream, s.flags, nullptr);
  }
  int kmax_ = 0;
  for (int i = 0; i < nk; ++i) kmax_ = std::max(kmax_, k_vals[i]);
  const int npl = l2min_col_planes(recall_mode(kmax_, n_total));
  RankArgs r{planes, n_local, nblk_pad, vtc_l2_sweep_row_block(), n_src, src_bounds, row_base, a_local, b_all, n_total, d, s.qn, s.qst, s.gmax,
             exact2_kappa(d), 0, nk, {0, 0, 0, 0}, (unsigned long long *)hits_a_from_b, s.flags, (int *)s.cand, s.cand_n, cdiv(n_local, RK_OW)};
  fill_k(r, k_vals, nk);
  launch_rank_one(r, npl, stream);
  VTC_LAUNCH_CHECK("l2_recall_shard_cols");
  return 0;
}

// diagnostics (tests/probes/sweep_v2_debug.py; not part of the public header): the raw block-minima planes of one distance GEMM.
// qb [nb, d], gb [na, d] bf16; qn, gn fp32 squared norms; rowk [4, ceil(na / 64), nb], colk [4, ceil(nb / rb), na] (or NULL)
extern "C" int vtc_debug_l2min(const void *qb, const void *gb, const float *qn, const float *gn, int nb, int na, int d, int rb,
                               unsigned *rowk, unsigned *colk, void *stream) {
  GemmEpi e;
  e.mode = EPI_L2MIN; e.out_dtype = VTC_F32; e.rown = qn; e.coln = gn;
  e.rowk = rowk; e.colk = colk; e.nblk_c = cdiv(na, 64); e.nblk_r = cdiv(nb, rb); e.rb = rb;
  return launch_gemm(qb, gb, nullptr, nullptr, nb, na, d, VTC_BF16, e, (hipStream_t)stream);
}

extern "C" size_t vtc_l2_topk_workspace_bytes(int n_gallery, int n_queries, int d, int precision, int rows_per_block) {
  const size_t v1 = plan(nullptr, n_gallery, n_queries, d, precision, rows_per_block).total;
  if (precision == VTC_SWEEP_EXACT && exact2_enabled(n_gallery, n_queries, 1))
    return std::max(v1, plan2(nullptr, n_gallery, n_queries, d, false).total);     // the depth decides at call time
  return v1;
}

// col_ids != nullptr: also the transposed direction (for every gallery row its nearest query rows), read off the same
// distance blocks by col_topk_kernel.
static int l2_topk_impl(const float *gallery, const float *queries, int ng, int nq, int d, int depth, int precision,
                        int64_t *ids, float *dists, const SweepWs &s, hipStream_t stream, int64_t *col_ids = nullptr,
                        float *col_dists = nullptr) {
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, stream, queries, s.qn, nq, d);
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(cdiv(ng, 4)), dim3(256), 0, stream, gallery, s.gn, ng, d);
  const int parts = precision == VTC_SWEEP_BF16X3 ? 3 : 1;
  if (precision != VTC_SWEEP_F32) {
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)(((size_t)nq * d + 255) / 256)), dim3(256), 0, stream, queries, s.qb, nq, d, parts, 0);
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)(((size_t)ng * d + 255) / 256)), dim3(256), 0, stream, gallery, s.gb, ng, d, parts, 1);
  }
  VTC_LAUNCH_CHECK("l2_topk prologue");
  for (int r0 = 0; r0 < nq; r0 += s.rows_per_block) {
    const int rows = min(s.rows_per_block, nq - r0);
    GemmEpi e;
    e.mode = EPI_L2DIST; e.out_dtype = VTC_F32; e.rown = s.qn + r0; e.coln = s.gn;
    int rc;
    if (precision == VTC_SWEEP_F32)
      rc = launch_gemm(queries + (size_t)r0 * d, gallery, nullptr, s.dist, rows, ng, d, VTC_F32, e, stream);
    else
      rc = launch_gemm(s.qb + (size_t)r0 * d * parts, s.gb, nullptr, s.dist, rows, ng, d * parts, VTC_BF16, e, stream);
    if (rc) return rc;
    {
      // enough (row, segment) waves to fill the chip (32 waves per CU)
      // (every segment pays its own warm-up insertions, so segments are used only when rows alone cannot fill the chip)
      int S = cdiv(8192, rows);
      S = S < 1 ? 1 : (S > MAX_SEG ? MAX_SEG : S);
      int seg_cols = cdiv(cdiv(ng, S), 1024) * 1024;
      S = cdiv(ng, seg_cols);
      ProfScope prof(VTC_PROF_TOPK, (double)rows * ng * 4, stream);
      hipLaunchKernelGGL(row_topk_kernel, dim3(cdiv(rows * S, 4)), dim3(256), 0, stream, s.dist, ng, rows, ng, depth, S, seg_cols, ids,
                         dists, (size_t)r0, s.part_d, s.part_i);
      if (S > 1)
        hipLaunchKernelGGL(topk_merge_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, s.part_d, s.part_i, rows, S, depth, ids, dists,
                           (size_t)r0);
    }
    VTC_LAUNCH_CHECK("row_topk");
    if (col_ids) {
      const int seg_rows = cdiv(cdiv(rows, s.c_seg), 64) * 64;
      ProfScope prof(VTC_PROF_TOPK, (double)rows * ng * 4, stream);
      hipLaunchKernelGGL(col_topk_kernel, dim3(cdiv(ng, 64) * s.c_seg), dim3(256), 0, stream, s.dist, ng, rows, ng, depth, r0, s.c_seg,
                         seg_rows, s.c_total, 0, r0 > 0 ? 1 : 0, s.cpart_d, s.cpart_i);
      VTC_LAUNCH_CHECK("col_topk");
    }
  }
  if (col_ids) {
    hipLaunchKernelGGL(topk_merge_kernel, dim3(cdiv(ng, 4)), dim3(256), 0, stream, s.cpart_d, s.cpart_i, ng, s.c_total, depth, col_ids,
                       col_dists, (size_t)0);
    VTC_LAUNCH_CHECK("col_topk merge");
  }
  return 0;
}

// EXACT tail: re-rank the candidate lists with fp64 distances, accept the rows whose list provably holds the true
// top-k, recompute the others by fp64 brute force.  qn / gn: fp32 squared norms of the queries / gallery rows.
static int exact_finish(const float *gallery, const float *queries, int ng, int nq, int d, int depth, int cdepth, const int64_t *cand,
                        const float *cand_d, const float *qn, const float *gn, float *gmax, int *flags, int64_t *ids, float *dists,
                        hipStream_t stream, const int *cand_n, const FallbackWs *fb, const Rescan *rs) {
  if (!cand_n) hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, stream, gn, ng, gmax);
  // split-bf16 candidate lists (cand_n == nullptr): worst-case error of a split-bf16 distance, relative to |q|^2 + max|g|^2.
  // With u = 2^-8 (bf16's unit roundoff): x = hi + lo + r, |lo| <= u |x|, |r| <= u^2 |x|; the GEMM forms hi.hi + hi.lo + lo.hi,
  // so the product misses lo.lo (u^2 |q||g|) and the two residual terms (2 u^2 |q||g|): 3 u^2 |q||g| <= 3 u^2 (|q|^2 + |g|^2) / 2,
  // twice that on the distance = 3 * 2^-16 (rounds 1-2 had 3 * 2^-18); fp32 accumulation of 3 d products (3 d * 2^-24), fp32 row
  // norms (d * 2^-24), the epilogue's three roundings.  (Block-minima lists arrive certified.)
  const float kappa = 3.0f / 65536.0f + 4.0f * d / 16777216.0f + 1e-6f;
  (void)hipMemsetAsync(flags, 0, sizeof(int), stream);
  const RerankArgs ra{queries, gallery, nq, ng, d, cand, cand_d, cdepth, depth, qn, gmax, kappa, ids, dists, flags, cand_n};
  {   // (work = bytes gathered at ~2 x depth candidates per row: the lists' lengths live on the device)
    ProfScope prof(VTC_PROF_TOPK, (double)nq * (2.0 * depth + 1.0) * d * 4, stream);
    hipLaunchKernelGGL(exact_rerank_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, stream, ra, ra, cdiv(nq, 4));
  }
  int f_first = 0;
  if (rs) {     // block-minima path: the uncertified owners are settled from their own planes (no pass over the gallery)
    ProfScope prof(VTC_PROF_TOPK, 0.0, stream);
    const RescanArgs rr{queries, gallery, ng, d, depth, flags, rs->keys, rs->R, rs->nblk, rs->bw, rs->nbs, rs->n_src, rs->src_base, rs->theta, ids, dists};
    hipLaunchKernelGGL(block_rescan_kernel, dim3(std::min(nq, 2048)), dim3(256), 0, stream, rr, rr, std::min(nq, 2048));
    VTC_LAUNCH_CHECK("l2_topk block rescan");
    return 0;
  }
  if (fb) {     // the first FB_ROWS uncertified rows: every row spread over FB_CHUNKS workgroups
    hipLaunchKernelGGL(exact_fallback_chunk_kernel, dim3(16, FB_CHUNKS), dim3(256), 0, stream, queries, gallery, ng, d, depth, flags, fb->part_d,
                       fb->part_i);
    hipLaunchKernelGGL(exact_fallback_merge_kernel, dim3(64), dim3(256), 0, stream, depth, flags, fb->part_d, fb->part_i, ids, dists);
    f_first = FB_ROWS;
  }
  hipLaunchKernelGGL(exact_fallback_kernel, dim3(std::min(nq, 2048)), dim3(256), 0, stream, queries, gallery, ng, d, depth, flags, ids,
                     dists, f_first);
  VTC_LAUNCH_CHECK("l2_topk exact");
  static const bool dbg = [] { const char *e = getenv("VTC_SWEEP_DEBUG"); return e && e[0] == '1'; }();
  if (dbg) {   // diagnostics only: synchronises
    int n_flagged = -1;
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(&n_flagged, flags, sizeof(int), hipMemcpyDeviceToHost);
    fprintf(stderr, "[sweep] exact_finish nq=%d ng=%d cdepth=%d %s: %d rows not certified -> fp64 brute force\n", nq, ng, cdepth,
            cand_n ? "block-minima" : "split-bf16", n_flagged);
  }
  return 0;
}

extern "C" int vtc_l2_topk(const float *gallery, const float *queries, int ng, int nq, int d, int depth, int precision,
                           int rows_per_block, int64_t *ids, float *dists, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(ng > 0 && nq > 0 && d > 0, "l2_topk: empty problem");
  VTC_CHECK(depth >= 1 && depth <= 64 && depth <= ng, "l2_topk: depth=%d must be in [1, min(64, n_gallery)]", depth);
  VTC_CHECK(precision >= VTC_SWEEP_F32 && precision <= VTC_SWEEP_EXACT, "l2_topk: bad precision %d", precision);
  VTC_CHECK(d % 64 == 0, "l2_topk: d=%d must be a multiple of 64", d);
  SweepWs s = plan((char *)ws, ng, nq, d, precision, rows_per_block);
  VTC_CHECK(ws && ws_bytes >= s.total, "l2_topk: workspace too small (%zu < %zu)", ws_bytes, s.total);
  if (precision != VTC_SWEEP_EXACT) return l2_topk_impl(gallery, queries, ng, nq, d, depth, precision, ids, dists, s, stream);
  if (exact2_enabled(ng, nq, depth) && rows_per_block <= 0) {
    Sweep2Ws s2 = plan2((char *)ws, ng, nq, d, false);
    VTC_CHECK(ws_bytes >= s2.total, "l2_topk: workspace too small (%zu < %zu)", ws_bytes, s2.total);
    return exact2_impl(gallery, queries, ng, nq, d, depth, ids, dists, nullptr, nullptr, s2, stream);
  }
  // EXACT: BF16X3 candidate lists, fp64 re-rank, verified superset property, fp64 brute force for the rest
  const int cdepth = exact_cdepth(depth, ng);
  if (int rc = l2_topk_impl(gallery, queries, ng, nq, d, cdepth, VTC_SWEEP_BF16X3, s.cand, s.cand_d, s, stream)) return rc;
  return exact_finish(gallery, queries, ng, nq, d, depth, cdepth, s.cand, s.cand_d, s.qn, s.gn, s.gmax, s.flags, ids, dists, stream, nullptr, nullptr);
}

// Both retrieval directions of RecallAtK (a -> b and b -> a, evaluation/eval.py:117-127) from ONE distance matrix
// D[i][j] = |b_i - a_j|^2: rows give, for every b row, its nearest a rows (ids_b2a: what vtc_l2_topk(gallery = a,
// queries = b) returns); columns give, for every a row, its nearest b rows (ids_a2b = vtc_l2_topk(gallery = b,
// queries = a)).  Same precisions; EXACT re-ranks both candidate sets in fp64.
extern "C" size_t vtc_l2_topk_bidir_workspace_bytes(int n_a, int n_b, int d, int precision, int rows_per_block) {
  const size_t v1 = plan(nullptr, n_a, n_b, d, precision, rows_per_block, true).total;
  if (precision == VTC_SWEEP_EXACT && exact2_enabled(n_a, n_b, 1) && exact2_enabled(n_b, n_a, 1))
    return std::max(v1, plan2(nullptr, n_a, n_b, d, true).total);
  return v1;
}

extern "C" int vtc_l2_topk_bidir(const float *a, const float *b, int n_a, int n_b, int d, int depth, int precision,
                                 int rows_per_block, int64_t *ids_b2a, float *dists_b2a, int64_t *ids_a2b, float *dists_a2b,
                                 void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(n_a > 0 && n_b > 0 && d > 0, "l2_topk_bidir: empty problem");
  VTC_CHECK(depth >= 1 && depth <= 64 && depth <= n_a && depth <= n_b, "l2_topk_bidir: depth=%d must be in [1, min(64, n_a, n_b)]", depth);
  VTC_CHECK(precision >= VTC_SWEEP_F32 && precision <= VTC_SWEEP_EXACT, "l2_topk_bidir: bad precision %d", precision);
  VTC_CHECK(d % 64 == 0, "l2_topk_bidir: d=%d must be a multiple of 64", d);
  VTC_CHECK(ids_b2a && ids_a2b, "l2_topk_bidir: both id outputs are required");
  SweepWs s = plan((char *)ws, n_a, n_b, d, precision, rows_per_block, true);
  VTC_CHECK(ws && ws_bytes >= s.total, "l2_topk_bidir: workspace too small (%zu < %zu)", ws_bytes, s.total);
  if (precision != VTC_SWEEP_EXACT)
    return l2_topk_impl(a, b, n_a, n_b, d, depth, precision, ids_b2a, dists_b2a, s, stream, ids_a2b, dists_a2b);
  if (exact2_enabled(n_a, n_b, depth) && exact2_enabled(n_b, n_a, depth) && rows_per_block <= 0) {
    Sweep2Ws s2 = plan2((char *)ws, n_a, n_b, d, true);
    VTC_CHECK(ws_bytes >= s2.total, "l2_topk_bidir: workspace too small (%zu < %zu)", ws_bytes, s2.total);
    return exact2_impl(a, b, n_a, n_b, d, depth, ids_b2a, dists_b2a, ids_a2b, dists_a2b, s2, stream);
  }
  const int cdepth = std::min(exact_cdepth(depth, n_a), exact_cdepth(depth, n_b));
  if (int rc = l2_topk_impl(a, b, n_a, n_b, d, cdepth, VTC_SWEEP_BF16X3, s.cand, s.cand_d, s, stream, s.cand2, s.cand2_d)) return rc;
  if (int rc = exact_finish(a, b, n_a, n_b, d, depth, cdepth, s.cand, s.cand_d, s.qn, s.gn, s.gmax, s.flags, ids_b2a, dists_b2a, stream, nullptr, nullptr))
    return rc;
  return exact_finish(b, a, n_b, n_a, d, depth, cdepth, s.cand2, s.cand2_d, s.gn, s.qn, s.qmax, s.flags, ids_a2b, dists_a2b, stream, nullptr, nullptr);
}

extern "C" int vtc_recall_hits(const int64_t *ids, int nq, int depth, int64_t target_offset, const int *k_vals, int nk,
                               long long *hits, void *stream) {
  VTC_CHECK(nk >= 1 && nk <= 4, "recall_hits: nk=%d must be in [1,4]", nk);
  int k[4] = {0, 0, 0, 0};
  for (int i = 0; i < nk; ++i) {
    VTC_CHECK(k_vals[i] >= 1 && k_vals[i] <= depth, "recall_hits: k=%d outside [1, depth=%d]", k_vals[i], depth);
    k[i] = k_vals[i];
  }
  {
    ProfScope prof(VTC_PROF_TOPK, (double)nq * depth * 8, (hipStream_t)stream);
    hipLaunchKernelGGL(recall_hits_kernel, dim3(cdiv(nq, 256)), dim3(256), 0, (hipStream_t)stream, ids, nq, depth, target_offset,
                       k[0], k[1], k[2], k[3], nk, (unsigned long long *)hits, (const int64_t *)nullptr, (unsigned long long *)nullptr);
  }
  VTC_LAUNCH_CHECK("recall_hits");
  return 0;
}

extern "C" int vtc_recall_hits_pair(const int64_t *ids_a, const int64_t *ids_b, int nq, int depth, int64_t target_offset, const int *k_vals,
                                    int nk, long long *hits_a, long long *hits_b, void *stream) {
  VTC_CHECK(ids_a && ids_b && hits_a && hits_b, "recall_hits_pair: null argument");
  VTC_CHECK(nk >= 1 && nk <= 4, "recall_hits_pair: nk=%d must be in [1,4]", nk);
  int k[4] = {0, 0, 0, 0};
  for (int i = 0; i < nk; ++i) {
    VTC_CHECK(k_vals[i] >= 1 && k_vals[i] <= depth, "recall_hits_pair: k=%d outside [1, depth=%d]", k_vals[i], depth);
    k[i] = k_vals[i];
  }
  {
    ProfScope prof(VTC_PROF_TOPK, 2.0 * nq * depth * 8, (hipStream_t)stream);
    hipLaunchKernelGGL(recall_hits_kernel, dim3(2 * cdiv(nq, 256)), dim3(256), 0, (hipStream_t)stream, ids_a, nq, depth, target_offset,
                       k[0], k[1], k[2], k[3], nk, (unsigned long long *)hits_a, ids_b, (unsigned long long *)hits_b);
  }
  VTC_LAUNCH_CHECK("recall_hits_pair");
  return 0;
}

extern "C" int vtc_l2_recall_bidir_supported(int n, int d) { return (n >= 1024 && d > 0 && d % 64 == 0) ? 1 : 0; }
extern "C" size_t vtc_l2_recall_bidir_workspace_bytes(int n, int d) { return plan2(nullptr, n, n, d, true).total; }
extern "C" int vtc_l2_recall_bidir(const float *a, const float *b, int n, int d, const int *k_vals, int nk, long long *hits_b_from_a,
                                   long long *hits_a_from_b, void *ws, size_t ws_bytes, void *stream) {
  VTC_CHECK(a && b && hits_b_from_a && hits_a_from_b && k_vals, "l2_recall_bidir: null argument");
  VTC_CHECK(vtc_l2_recall_bidir_supported(n, d), "l2_recall_bidir: n=%d must be >= 1024 and d=%d a multiple of 64 (else: vtc_l2_topk + vtc_recall_hits)", n, d);
  VTC_CHECK(nk >= 1 && nk <= 4, "l2_recall_bidir: nk=%d must be in [1,4]", nk);
  for (int i = 0; i < nk; ++i) VTC_CHECK(k_vals[i] >= 1 && k_vals[i] <= n, "l2_recall_bidir: k=%d outside [1, n=%d]", k_vals[i], n);
  Sweep2Ws s = plan2((char *)ws, n, n, d, true);
  VTC_CHECK(ws && ws_bytes >= s.total, "l2_recall_bidir: workspace too small (%zu < %zu)", ws_bytes, s.total);
  return recall_bidir_impl(a, b, n, d, k_vals, nk, (unsigned long long *)hits_b_from_a, (unsigned long long *)hits_a_from_b, s, (hipStream_t)stream);
}

extern "C" int vtc_similarity(const float *v, const float *t, int nv, int nt, int d, const float *logit_scale, float *sim,
                              void *stream) {
  GemmEpi e;
  e.mode = EPI_SCALE; e.out_dtype = VTC_F32; e.scale_log = logit_scale;
  return launch_gemm(v, t, nullptr, sim, nv, nt, d, VTC_F32, e, (hipStream_t)stream);
}

extern "C" size_t vtc_clip_loss_workspace_bytes(int n) { return (size_t)2 * n * sizeof(float); }

extern "C" int vtc_clip_loss(const float *sim, int n, float *loss, void *ws, size_t ws_bytes, void *stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  VTC_CHECK(n > 0, "clip_loss: n=%d", n);
  VTC_CHECK(ws && ws_bytes >= vtc_clip_loss_workspace_bytes(n), "clip_loss: workspace too small");
  hipLaunchKernelGGL(lse_terms_kernel, dim3(cdiv(2 * n, 4)), dim3(256), 0, stream, sim, n, (float *)ws);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, stream, (const float *)ws, n, loss);
  VTC_LAUNCH_CHECK("clip_loss");
  return 0;
}
