// attention.hip -- softmax(q k^T / sqrt(64) [+ causal]) v for the short sequences of this path:
//   time attention   L = F (8/16)     model/timesformer_clip_alt.py:142-147
//   space attention  L = 1 + 49       model/timesformer_clip_alt.py:152-158 (cls replicated per frame)
//   ViT              L = 50           upstream VisionTransformer
//   text             L = 77 causal    upstream CLIP.encode_text (mask = full(-inf).triu_(1))
//   CAM              L = 1 + ncomms   model/model.py:155
// Replaces `attn` / `multi_head_attention` (model/timesformer_clip_alt.py:36-67) minus the two
// projections, which are GEMMs.
//
// gfx950 design: ONE WAVE per (sequence, head); the whole sequence lives in that wave.
//   * the einops rearranges of the reference ("(b h w) t", "(b t) (h w)", cls replicate) are pure
//     index arithmetic here: token p of sequence s is fetched from its row of the [rows, 3W]
//     qkv buffer through an affine row map, and written back through the same map;
//   * S^T = K Q^T on the matrix cores (keys on the accumulator rows, queries on the lanes), so
//     the softmax over keys is lane-local + two xor-shuffles, and the probability accumulators
//     ARE the B operand of the P.V product -- no LDS round trip, no transposition of P;
//   * K fragments are loaded once from global into registers and reused for every query tile;
//     V is staged once per (sequence, head) into LDS transposed ([d][key]) so the P.V "A" operand
//     is a contiguous 8/16-byte LDS read;
//   * bf16: v_mfma_f32_16x16x32_bf16 (P rounded to bf16 for P.V); fp32: v_mfma_f32_16x16x4_f32.
// head_dim is fixed at 64 (upstream: heads = width / 64).
#include "common.h"

namespace {

struct AttnParams {
  const char *qkv;
  char *out;
  float *cls_out;
  int n_seq, L, heads, causal;
  int s2, a0, a1, a2, a3, pstride;
  const int *seq_offsets;   // ragged mode: sequence s = rows [seq_offsets[s], seq_offsets[s+1]) (overrides the affine map)
  int W;  // model width = heads * 64
};

template <typename T> struct AT;
template <> struct AT<bf16_t> {
  static constexpr int NCH = 8;   // 16-byte chunks per 64-element head row
  static constexpr int KS = 2;    // QK k-steps (4 chunks each)
  static constexpr int EPC = 8;   // elements per chunk
};
template <> struct AT<f16_t> : AT<bf16_t> {};
template <> struct AT<float> {
  static constexpr int NCH = 16;
  static constexpr int KS = 4;
  static constexpr int EPC = 4;
};

__device__ __forceinline__ void mma_qk(bf16_t, const uint4 &k, const uint4 &q, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, k), __builtin_bit_cast(bf16x8, q), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_qk(f16_t, const uint4 &k, const uint4 &q, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, k), __builtin_bit_cast(f16x8, q), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_qk(float, const uint4 &k, const uint4 &q, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, k.x), __builtin_bit_cast(float, q.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, k.y), __builtin_bit_cast(float, q.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, k.z), __builtin_bit_cast(float, q.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, k.w), __builtin_bit_cast(float, q.w), acc, 0, 0, 0);
}

template <typename T, int NT>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && NT <= 4) ? 4 : 1) void attn_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int SZ = sizeof(T);
  constexpr int VS = 16 * NT + 4;           // Vt row stride in elements (keys), keeps 8/16-byte alignment
  constexpr int NCH = AT<T>::NCH, KS = AT<T>::KS, EPC = AT<T>::EPC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  T *vt = reinterpret_cast<T *>(lds_raw) + (size_t)wave * 64 * VS;

  int gw = blockIdx.x * 4 + wave;
  const int total = p.n_seq * p.heads;
  const bool active = gw < total;
  if (!active) gw = total - 1;              // keep every wave alive for the barrier; stores are predicated
  const int s = gw / p.heads, h = gw - s * p.heads;
  const int s_hi = s / p.s2, s_lo = s - s_hi * p.s2;
  const long base = p.seq_offsets ? (long)p.seq_offsets[s] : (long)s_hi * p.a1 + (long)s_lo * p.a2 + p.a0;
  const int first = 1 + s_lo * p.a3;
  const int L = p.seq_offsets ? p.seq_offsets[s + 1] - p.seq_offsets[s] : p.L;
  const size_t ld = (size_t)3 * p.W * SZ;   // qkv row bytes
  auto row_of = [&](int tok) -> long { return tok == 0 ? base : base + first + (long)(tok - 1) * p.pstride; };

  // ---- V -> LDS (transposed) and K fragments -> registers ------------------------------------
  // Every load of a batch is requested before the first one is used, and the K fragments with the first batch: a V row past the
  // sequence is clamped to the last row and zeroed afterwards, not skipped -- a branch per load made hipcc wait for each load in
  // turn (`if (tok < L) raw = load` x 8: nine dependent global round trips per (sequence, head) with the K fragments behind them,
  // ~10 of the ~17 us an item took; profiles/r04_experiments.txt 19).
  uint4 kf[NT][KS];
  {
    const char *vbase = p.qkv + (size_t)(2 * p.W + h * 64) * SZ;
    const char *kbase = p.qkv + (size_t)(p.W + h * 64) * SZ;
    constexpr int NIT = (16 * NT * NCH) / 64, VB = NIT < 8 ? NIT : 8;       // loads per lane, per batch (8 x 16 B = 32 registers)
#pragma unroll
    for (int b0 = 0; b0 < NIT; b0 += VB) {
      uint4 raw[VB];
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        if (b0 + u >= NIT) continue;                      // (the last batch may be short: compile-time after unrolling)
        const int idx = (b0 + u) * 64 + lane;
        const int tok = idx / NCH, ch = idx - tok * NCH;
        raw[u] = *reinterpret_cast<const uint4 *>(vbase + (size_t)row_of(min(tok, L - 1)) * ld + ch * 16);
      }
      if (b0 == 0) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          const int tok = min(kt * 16 + c16, L - 1);
          const char *r = kbase + (size_t)row_of(tok) * ld;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) kf[kt][ks] = *reinterpret_cast<const uint4 *>(r + (4 * ks + g) * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        if (b0 + u >= NIT) continue;
        const int idx = (b0 + u) * 64 + lane;
        const int tok = idx / NCH, ch = idx - tok * NCH;
        const uint4 v = tok < L ? raw[u] : make_uint4(0, 0, 0, 0);
        const T *e = reinterpret_cast<const T *>(&v);
#pragma unroll
        for (int j = 0; j < EPC; ++j) vt[(ch * EPC + j) * VS + tok] = e[j];
      }
    }
  }
  __syncthreads();  // Vt visible to every lane of the wave (waves do not share LDS regions)

  const char *qbase = p.qkv + (size_t)(h * 64) * SZ;
  // query fragments are fetched one tile ahead: the loads of tile qt + 1 are in flight while tile qt is multiplied,
  // soft-maxed and stored (the compiler cannot hoist them itself across the stores to p.out)
  auto load_q = [&](int qt, uint4 (&q)[KS]) {
    const char *r = qbase + (size_t)row_of(min(qt * 16 + c16, L - 1)) * ld;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) q[ks] = *reinterpret_cast<const uint4 *>(r + (4 * ks + g) * 16);
  };
  uint4 qf[KS], qn[KS];
  load_q(0, qf);
  for (int qt = 0; qt < NT; ++qt) {
    if (qt * 16 >= L) break;
    const int qtok = qt * 16 + c16;
    if (qt + 1 < NT && (qt + 1) * 16 < L) load_q(qt + 1, qn);
    f32x4 sc[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if ((p.causal && kt > qt) || kt * 16 >= L) {   // tile entirely in the future of every query, or past the sequence end
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[kt][r] = -INFINITY;
        continue;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) mma_qk(T(), kf[kt][ks], qf[ks], sc[kt]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + g * 4 + r;
        float v = sc[kt][r] * 0.125f;       // q * head_dim^-0.5 (timesformer_clip_alt.py:48,52); exact power of two
        if (key >= L || (p.causal && key > qtok)) v = -INFINITY;
        sc[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = sizeof(T) == 4 ? expf(sc[kt][r] - mx) : __expf(sc[kt][r] - mx);
        sc[kt][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // ---- O^T[d][query] = sum_key Vt[d][key] * P[query][key] ---------------------------------
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kk = 0; kk < (NT + 1) / 2; ++kk) {
        const int k0 = 2 * kk, k1 = 2 * kk + 1;
        bf16x8 pf;   // raw 16-bit lanes of the operand format (bf16 or IEEE half)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[r] = (short)cvt16<T>(sc[k0][r]);
          pf[4 + r] = k1 < NT ? (short)cvt16<T>(sc[k1 < NT ? k1 : k0][r]) : (short)0;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned short *vr = reinterpret_cast<const unsigned short *>(vt) + (dt * 16 + c16) * VS + g * 4;
          const uint2 lo = *reinterpret_cast<const uint2 *>(vr + k0 * 16);
          uint2 hi = make_uint2(0, 0);
          if (k1 < NT) hi = *reinterpret_cast<const uint2 *>(vr + k1 * 16);
          const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
          mma_qk(T(), vf, __builtin_bit_cast(uint4, pf), o[dt]);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const float4 vf = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(vt) + (dt * 16 + c16) * VS + kt * 16 + g * 4);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, sc[kt][0], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, sc[kt][1], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, sc[kt][2], o[dt], 0, 0, 0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, sc[kt][3], o[dt], 0, 0, 0);
        }
      }
    }

    // ---- store: lane holds O[query = qtok][d = 16 dt + 4 g .. +3] ---------------------------
    if (active && qtok < L) {
      if (p.cls_out && qtok == 0) {
        float *dst = p.cls_out + (size_t)s * p.W + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<float4 *>(dst + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      } else {
        T *dst = reinterpret_cast<T *>(p.out) + (size_t)row_of(qtok) * p.W + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          ElemOps<T>::store4(dst + dt * 16, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = qn[ks];
  }
}

// ---- longer sequences: 80 < L <= 272 (ViT-B/16: 1 + 196 tokens per frame, ViT-L/14: 1 + 256; model/timesformer_clip_alt.py:297-310) ----
// ONE WORKGROUP (4 waves) per (sequence, head): the K rows and V^T of the sequence are staged once into LDS by all four waves,
// then wave w takes the query tiles w, w + 4, ...; per query tile an ONLINE softmax over chunks of 64 keys (4 MFMA key tiles):
// running maximum m and running sum l per query (lane-local: queries sit on lanes, as in attn_kernel), the P.V accumulators
// rescaled by exp(m_old - m_new) when the maximum moves.  Same fragment maps, scaling, masking and store as attn_kernel.
template <typename T>
__global__ __launch_bounds__(256) void attn_tiled_kernel(const AttnParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  constexpr int SZ = sizeof(T);
  constexpr int NCH = AT<T>::NCH, KS = AT<T>::KS, EPC = AT<T>::EPC;
  constexpr int KROW = 64 * SZ + 16;        // K row stride in bytes (16-byte pad: spreads the 16 rows of a fragment over the banks)
  const int Lp = 16 * ntiles, VS = Lp + 4;  // V^T row stride in elements (keeps 8 / 16-byte alignment)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c16 = lane & 15;
  char *kl = lds_raw;
  T *vt = reinterpret_cast<T *>(lds_raw + (size_t)Lp * KROW);

  const int gw = blockIdx.x;
  const int s = gw / p.heads, h = gw - s * p.heads;
  const int s_hi = s / p.s2, s_lo = s - s_hi * p.s2;
  const long base = p.seq_offsets ? (long)p.seq_offsets[s] : (long)s_hi * p.a1 + (long)s_lo * p.a2 + p.a0;
  const int first = 1 + s_lo * p.a3;
  const int L = p.seq_offsets ? p.seq_offsets[s + 1] - p.seq_offsets[s] : p.L;
  const size_t ld = (size_t)3 * p.W * SZ;
  auto row_of = [&](int tok) -> long { return tok == 0 ? base : base + first + (long)(tok - 1) * p.pstride; };

  // ---- K rows and V^T -> LDS (rows past the sequence end: zeros; they are masked below) ----
  {
    const char *kbase = p.qkv + (size_t)(p.W + h * 64) * SZ, *vbase = p.qkv + (size_t)(2 * p.W + h * 64) * SZ;
    for (int idx = tid; idx < Lp * NCH; idx += 256) {
      const int tok = idx / NCH, ch = idx - tok * NCH;
      uint4 kr = make_uint4(0, 0, 0, 0), vr = make_uint4(0, 0, 0, 0);
      if (tok < L) {
        const size_t ro = (size_t)row_of(tok) * ld + ch * 16;
        kr = *reinterpret_cast<const uint4 *>(kbase + ro);
        vr = *reinterpret_cast<const uint4 *>(vbase + ro);
      }
      *reinterpret_cast<uint4 *>(kl + (size_t)tok * KROW + ch * 16) = kr;
      const T *e = reinterpret_cast<const T *>(&vr);
#pragma unroll
      for (int j = 0; j < EPC; ++j) vt[(ch * EPC + j) * VS + tok] = e[j];
    }
  }
  __syncthreads();

  const char *qbase = p.qkv + (size_t)(h * 64) * SZ;
  for (int qt = wave; qt < ntiles && qt * 16 < L; qt += 4) {
    const int qtok = qt * 16 + c16;
    uint4 qf[KS];
    {
      const char *r = qbase + (size_t)row_of(min(qtok, L - 1)) * ld;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4 *>(r + (4 * ks + g) * 16);
    }
    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < ntiles; kc += 4) {
      if (kc * 16 >= L || (p.causal && kc > qt)) break;       // the rest lies past the sequence end / in every query's future
      f32x4 sc[4];
      float cm = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kt = kc + j;
        sc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kt >= ntiles || kt * 16 >= L || (p.causal && kt > qt)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[j][r] = -INFINITY;
          continue;
        }
        const char *kr = kl + (size_t)(kt * 16 + c16) * KROW + g * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mma_qk(T(), *reinterpret_cast<const uint4 *>(kr + ks * 64), qf[ks], sc[j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + g * 4 + r;
          float v = sc[j][r] * 0.125f;       // q * head_dim^-0.5 (timesformer_clip_alt.py:48,52)
          if (key >= L || (p.causal && key > qtok)) v = -INFINITY;
          sc[j][r] = v;
          cm = fmaxf(cm, v);
        }
      }
      cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
      cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
      const float mn = fmaxf(m, cm);          // finite from the first chunk on: key 0 is visible to every query
      const float alpha = sizeof(T) == 4 ? expf(m - mn) : __expf(m - mn);
      m = mn;
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = sizeof(T) == 4 ? expf(sc[j][r] - mn) : __expf(sc[j][r] - mn);
          sc[j][r] = e;
          ps += e;
        }
      l = l * alpha + ps;                      // this lane's keys only; summed over the four key groups at the end
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
      // O^T[d][query] += sum_key Vt[d][key] * P[query][key]
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int k0 = kc + 2 * jj, k1 = k0 + 1;
          if (k0 >= ntiles) break;
          const bool has1 = k1 < ntiles;
          bf16x8 pf;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pf[r] = (short)cvt16<T>(sc[2 * jj][r]);
            pf[4 + r] = has1 ? (short)cvt16<T>(sc[2 * jj + 1][r]) : (short)0;
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const unsigned short *vr = reinterpret_cast<const unsigned short *>(vt) + (dt * 16 + c16) * VS + g * 4;
            const uint2 lo = *reinterpret_cast<const uint2 *>(vr + k0 * 16);
            uint2 hi = make_uint2(0, 0);
            if (has1) hi = *reinterpret_cast<const uint2 *>(vr + k1 * 16);
            mma_qk(T(), make_uint4(lo.x, lo.y, hi.x, hi.y), __builtin_bit_cast(uint4, pf), o[dt]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kt = kc + j;
          if (kt >= ntiles) break;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const float4 vf = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(vt) + (dt * 16 + c16) * VS + kt * 16 + g * 4);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, sc[j][0], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, sc[j][1], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, sc[j][2], o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, sc[j][3], o[dt], 0, 0, 0);
          }
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    if (qtok < L) {
      if (p.cls_out && qtok == 0) {
        float *dst = p.cls_out + (size_t)s * p.W + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<float4 *>(dst + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      } else {
        T *dst = reinterpret_cast<T *>(p.out) + (size_t)row_of(qtok) * p.W + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          ElemOps<T>::store4(dst + dt * 16, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
      }
    }
  }
}

// ---- any head_dim, short contiguous sequences (round 4): the Context Adapter Module when feature_dim / n_heads != 64 --------------
// (ViT-L/14: 768-d features with the reference's default n_heads = 8 -> head_dim 96, model/model.py:396-398).  L <= 16 tokens,
// head_dim <= 128: one wave per (sequence, head), lane l owns head dimensions l and l + 64; the L x L scores are wave reductions
// (the CAM is 1 + nc = 6 tokens: 36 of them), softmax in registers, q scaled by head_dim^-0.5 as upstream nn.MultiheadAttention
// does.  Not a throughput kernel: the module is ~0.06 % of a forward's FLOPs.
template <typename T>
__global__ __launch_bounds__(256) void attn_generic_small_kernel(const T *__restrict__ qkv, T *__restrict__ out, int n_seq, int L, int heads, int hd) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= n_seq * heads) return;
  const int s = gw / heads, h = gw - s * heads;
  const int W = heads * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  const bool d0 = lane < hd, d1 = lane + 64 < hd;
  float q[16][2], k[16][2], v[16][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    q[i][0] = q[i][1] = k[i][0] = k[i][1] = v[i][0] = v[i][1] = 0.f;
    if (i < L) {
      const T *r = qkv + (size_t)(s * L + i) * 3 * W + h * hd;
      if (d0) { q[i][0] = ElemOps<T>::load(r + lane) * scale; k[i][0] = ElemOps<T>::load(r + W + lane); v[i][0] = ElemOps<T>::load(r + 2 * W + lane); }
      if (d1) { q[i][1] = ElemOps<T>::load(r + lane + 64) * scale; k[i][1] = ElemOps<T>::load(r + W + lane + 64); v[i][1] = ElemOps<T>::load(r + 2 * W + lane + 64); }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < L) {                                   // wave-uniform (no `break`: the loop stays fully unrolled, q / k / v in registers)
    float sc[16], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      sc[j] = -INFINITY;
      if (j < L) {
        sc[j] = wave_sum(q[i][0] * k[j][0] + q[i][1] * k[j][1]);
        mx = fmaxf(mx, sc[j]);
      }
    }
    float sum = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < L) {
        const float e = expf(sc[j] - mx);
        sum += e;
        o0 = fmaf(e, v[j][0], o0);
        o1 = fmaf(e, v[j][1], o1);
      }
    }
    T *dst = out + (size_t)(s * L + i) * W + h * hd;
    if (d0) ElemOps<T>::store(dst + lane, o0 / sum);
    if (d1) ElemOps<T>::store(dst + lane + 64, o1 / sum);
    }
  }
}

// Global cls attention of model/timesformer_clip.py:81,158: the cls query of each item attends to ALL
// T tokens of the item.  One wave per (item, head); lanes = the 64 head dimensions; scores via a
// wave reduction per key, kept in LDS; output row = the item's cls row.  T up to 1024.
template <typename T>
__global__ __launch_bounds__(256) void cls_global_attn_kernel(const T *__restrict__ qkv, T *__restrict__ out, int n_items, int Ttok,
                                                              int heads) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *sc = reinterpret_cast<float *>(lds_raw) + (size_t)wave * 1024;
  int gw = blockIdx.x * 4 + wave;
  const int total = n_items * heads;
  const bool active = gw < total;
  if (!active) gw = total - 1;
  const int item = gw / heads, h = gw - item * heads;
  const int W = heads * 64;
  const T *base = qkv + (size_t)item * Ttok * 3 * W;
  const float q = ElemOps<T>::load(base + h * 64 + lane) * 0.125f;
  float mx = -INFINITY;
  for (int j = 0; j < Ttok; ++j) {
    const float s = wave_sum(q * ElemOps<T>::load(base + (size_t)j * 3 * W + W + h * 64 + lane));
    if (lane == (j & 63)) sc[j] = s;
    mx = fmaxf(mx, s);
  }
  __syncthreads();
  float sum = 0.f;
  for (int j = lane; j < Ttok; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  __syncthreads();
  float o = 0.f;
  for (int j = 0; j < Ttok; ++j) o += sc[j] * ElemOps<T>::load(base + (size_t)j * 3 * W + 2 * W + h * 64 + lane);
  if (active) ElemOps<T>::store(out + (size_t)item * Ttok * W + h * 64 + lane, o / sum);
}

// ---- one query per sequence: the LAST block of a tower, where only the row that reaches the output asks (towers.hip, last_block_tail) ----
// Sequence o attends with the ONE projected query q[o / s2] (compact [n_q, W], operand format) over its keys and values in the packed
// qkv buffer (the Q third of qkv is never read -- for this block it is never computed); out[o] [W] fp32.  Key rows: the affine map of
// launch_attention, or (eot != NULL) the text tower's rows base .. eot[o] with base = offs ? offs[o] : o ctx (the EOT query sees its
// whole causal prefix).  One wave per (sequence, head).  A 64-lane pass covers 64 / NCH keys x NCH 16-byte chunks of the head's 64
// dimensions: scores = 8-element partial dots reduced over a key's NCH lanes, probabilities through LDS, P.V accumulated per chunk and
// reduced over the passes' key slots at the end.  fp32 throughout (the all-rows kernel rounds P to the operand format for its MFMAs).
struct SqParams {
  const char *qkv, *q;
  float *out;
  int n_out, heads, W;
  int L, s2, a0, a1, a2, a3, pstride;
  const int *eot, *offs;
  int ctx;
};

template <typename T>
__device__ __forceinline__ void unpack_chunk(const uint4 &raw, float (&f)[AT<T>::EPC]) {
  if constexpr (sizeof(T) == 4) {
    f[0] = __uint_as_float(raw.x); f[1] = __uint_as_float(raw.y); f[2] = __uint_as_float(raw.z); f[3] = __uint_as_float(raw.w);
  } else {
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = up16<T>((unsigned short)(w[e] & 0xFFFFu));
      f[2 * e + 1] = up16<T>((unsigned short)(w[e] >> 16));
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sq_attn_kernel(const SqParams p) {
  constexpr int SZ = sizeof(T), NCH = AT<T>::NCH, EPC = AT<T>::EPC, KP = 64 / NCH;
  constexpr int SQ_MAXL = 320, SQ_PL = SQ_MAXL / 64;     // keys per sequence (ViT-L/14: 257), keys per lane
  __shared__ float pl[4][SQ_MAXL];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * 4 + wave;
  if (gw >= p.n_out * p.heads) return;              // no workgroup barrier below: waves are independent
  const int o = gw / p.heads, h = gw - o * p.heads;
  long base;
  int first, pstride, L, qi;
  if (p.eot) {
    base = p.offs ? (long)p.offs[o] : (long)o * p.ctx;
    L = p.eot[o] - (int)base + 1;
    first = 1; pstride = 1; qi = o;
  } else {
    const int s_hi = o / p.s2, s_lo = o - s_hi * p.s2;
    base = (long)s_hi * p.a1 + (long)s_lo * p.a2 + p.a0;
    first = 1 + s_lo * p.a3; pstride = p.pstride; L = p.L; qi = s_hi;
  }
  L = min(max(L, 1), SQ_MAXL);
  auto row_of = [&](int tok) -> long { return tok == 0 ? base : base + first + (long)(tok - 1) * pstride; };
  const size_t ld = (size_t)3 * p.W * SZ;
  const int ch = lane % NCH, kq = lane / NCH;
  float qf[EPC];
  unpack_chunk<T>(*reinterpret_cast<const uint4 *>(p.q + ((size_t)qi * p.W + h * 64) * SZ + ch * 16), qf);
  float *pw = pl[wave];
  // scores
  const char *kbase = p.qkv + (size_t)(p.W + h * 64) * SZ + ch * 16;
  for (int k0 = 0; k0 < L; k0 += KP) {
    const int tok = k0 + kq;
    float d = 0.f;
    if (tok < L) {
      float kf[EPC];
      unpack_chunk<T>(*reinterpret_cast<const uint4 *>(kbase + (size_t)row_of(tok) * ld), kf);
#pragma unroll
      for (int e = 0; e < EPC; ++e) d = fmaf(qf[e], kf[e], d);
    }
#pragma unroll
    for (int x = 1; x < NCH; x <<= 1) d += __shfl_xor(d, x, 64);
    if (ch == 0 && tok < L) pw[tok] = d * 0.125f;       // q * head_dim^-0.5 (timesformer_clip_alt.py:48,52)
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // softmax over the L keys (SQ_PL per lane)
  float sv[SQ_PL], mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SQ_PL; ++i) {
    sv[i] = lane + 64 * i < L ? pw[lane + 64 * i] : -INFINITY;
    mx = fmaxf(mx, sv[i]);
  }
  mx = wave_max(mx);
  float es = 0.f;
#pragma unroll
  for (int i = 0; i < SQ_PL; ++i) {
    sv[i] = lane + 64 * i < L ? __expf(sv[i] - mx) : 0.f;
    es += sv[i];
  }
  const float inv = 1.0f / wave_sum(es);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < SQ_PL; ++i) pw[lane + 64 * i] = sv[i] * inv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // out[d] = sum_key P[key] V[key][d]
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  const char *vbase = p.qkv + (size_t)(2 * p.W + h * 64) * SZ + ch * 16;
#pragma unroll 4
  for (int k0 = 0; k0 < L; k0 += KP) {
    const int tok = k0 + kq;
    if (tok < L) {
      float vf[EPC];
      unpack_chunk<T>(*reinterpret_cast<const uint4 *>(vbase + (size_t)row_of(tok) * ld), vf);
      const float pk = pw[tok];
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] = fmaf(pk, vf[e], acc[e]);
    }
  }
#pragma unroll
  for (int x = NCH; x < 64; x <<= 1)
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] += __shfl_xor(acc[e], x, 64);
  if (kq == 0) {
    float *dst = p.out + (size_t)o * p.W + h * 64 + ch * EPC;
    *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if constexpr (EPC == 8) *reinterpret_cast<float4 *>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// out[i] = mean_f x[i F + f] in the operand format (F = 1: a conversion)
template <typename T>
__global__ __launch_bounds__(256) void mean_cast_kernel(const float *__restrict__ x, T *__restrict__ out, int n, int F, int W) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * W) return;
  const size_t i = idx / W, c = idx - i * W;
  float s = 0.f;
  for (int f = 0; f < F; ++f) s += x[(i * F + f) * W + c];
  s /= (float)F;
  if constexpr (sizeof(T) == 4) out[idx] = s;
  else reinterpret_cast<unsigned short *>(out)[idx] = cvt16<T>(s);
}

template <typename T, int NT>
int run(const AttnParams &p, hipStream_t stream) {
  constexpr int VS = 16 * NT + 4;
  const size_t shmem = (size_t)4 * 64 * VS * sizeof(T);
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&attn_kernel<T, NT>), (int)shmem, "attention")) return 1;
  const int total = p.n_seq * p.heads;
  hipLaunchKernelGGL((attn_kernel<T, NT>), dim3(cdiv(total, 4)), dim3(256), shmem, stream, p);
  VTC_LAUNCH_CHECK("attention");
  return 0;
}

constexpr int ATTN_TILED_MAX_L = 272;
template <typename T>
int run_tiled(const AttnParams &p, hipStream_t stream) {
  VTC_CHECK(p.L <= ATTN_TILED_MAX_L, "attention: sequence length %d > %d unsupported", p.L, ATTN_TILED_MAX_L);
  const int ntiles = cdiv(p.L, 16), Lp = 16 * ntiles;
  const size_t shmem = (size_t)Lp * (64 * sizeof(T) + 16) + (size_t)64 * (Lp + 4) * sizeof(T);     // K rows + V^T: 74 KB (16-bit) / 145 KB (fp32) at L = 272
  static PerDeviceOnce attr;
  if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&attn_tiled_kernel<T>), 160 * 1024, "attention_tiled")) return 1;
  hipLaunchKernelGGL((attn_tiled_kernel<T>), dim3(p.n_seq * p.heads), dim3(256), shmem, stream, p, ntiles);
  VTC_LAUNCH_CHECK("attention_tiled");
  return 0;
}

template <typename T>
int dispatch(const AttnParams &p, hipStream_t stream) {
  switch (cdiv(p.L, 16)) {
    case 1: return run<T, 1>(p, stream);
    case 2: return run<T, 2>(p, stream);
    case 3: return run<T, 3>(p, stream);
    case 4: return run<T, 4>(p, stream);
    case 5: return run<T, 5>(p, stream);
  }
  return run_tiled<T>(p, stream);
}

}  // namespace

int launch_attention(const void *qkv, void *out, float *cls_out, int n_seq, int L, int heads, int causal, int s2,
                     int a0, int a1, int a2, int a3, int pstride, int dtype, hipStream_t stream) {
  VTC_CHECK(n_seq > 0 && L > 0 && heads > 0 && s2 > 0, "attention: bad sizes n_seq=%d L=%d heads=%d s2=%d", n_seq, L, heads, s2);
  AttnParams p;
  p.qkv = (const char *)qkv; p.out = (char *)out; p.cls_out = cls_out;
  p.n_seq = n_seq; p.L = L; p.heads = heads; p.causal = causal;
  p.s2 = s2; p.a0 = a0; p.a1 = a1; p.a2 = a2; p.a3 = a3; p.pstride = pstride;
  p.seq_offsets = nullptr;
  p.W = heads * 64;
  ProfScope prof(VTC_PROF_ATTN, 4.0 * L * L * 64 * (double)n_seq * heads, stream);
  if (dtype == VTC_F16) return dispatch<f16_t>(p, stream);
  return dtype == VTC_BF16 ? dispatch<bf16_t>(p, stream) : dispatch<float>(p, stream);
}

// contiguous sequences of L <= 16 tokens, any head_dim <= 128 (the CAM with head_dim != 64); unmasked
int launch_attention_generic_small(const void *qkv, void *out, int n_seq, int L, int heads, int hd, int dtype, hipStream_t stream) {
  VTC_CHECK(n_seq > 0 && L >= 1 && L <= 16 && heads > 0 && hd >= 1 && hd <= 128, "attention (generic head_dim): n_seq=%d L=%d heads=%d head_dim=%d", n_seq, L, heads, hd);
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16, "attention (generic head_dim): dtype %d", dtype);
  ProfScope prof(VTC_PROF_ATTN, 4.0 * L * L * hd * (double)n_seq * heads, stream);
  const dim3 grid(cdiv(n_seq * heads, 4));
  if (dtype == VTC_F32) hipLaunchKernelGGL((attn_generic_small_kernel<float>), grid, dim3(256), 0, stream, (const float *)qkv, (float *)out, n_seq, L, heads, hd);
  else hipLaunchKernelGGL((attn_generic_small_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t *)qkv, (bf16_t *)out, n_seq, L, heads, hd);
  VTC_LAUNCH_CHECK("attention_generic_small");
  return 0;
}

// Ragged batch: sequence s occupies the packed rows [seq_offsets[s], seq_offsets[s+1]); max_L bounds the lengths.
int launch_attention_ragged(const void *qkv, void *out, int n_seq, int max_L, int heads, int causal, const int *seq_offsets,
                            double flops, const int *rows_dev, int dtype, hipStream_t stream) {
  VTC_CHECK(n_seq > 0 && max_L > 0 && heads > 0 && seq_offsets, "attention_ragged: bad arguments");
  AttnParams p;
  p.qkv = (const char *)qkv; p.out = (char *)out; p.cls_out = nullptr;
  p.n_seq = n_seq; p.L = max_L; p.heads = heads; p.causal = causal;
  p.s2 = 1; p.a0 = 0; p.a1 = 0; p.a2 = 0; p.a3 = 0; p.pstride = 1;
  p.seq_offsets = seq_offsets;
  p.W = heads * 64;
  ProfScope prof(VTC_PROF_ATTN, flops, stream, rows_dev);      // rows_dev: `flops` is per row
  if (dtype == VTC_F16) return dispatch<f16_t>(p, stream);
  return dtype == VTC_BF16 ? dispatch<bf16_t>(p, stream) : dispatch<float>(p, stream);
}

// One query per sequence (struct SqParams): affine key map as launch_attention (eot == NULL) or the text tower's EOT query.
int launch_single_query_attention(const void *qkv, const void *q, float *out, int n_out, int L, int heads, int s2, int a0, int a1, int a2,
                                  int a3, int pstride, const int *eot, const int *offs, int ctx, int dtype, hipStream_t stream) {
  VTC_CHECK(n_out > 0 && heads > 0 && s2 > 0 && (eot || (L > 0 && L <= 320)), "single_query_attention: bad sizes n_out=%d L=%d", n_out, L);
  VTC_CHECK(!eot || ctx <= 320, "single_query_attention: context %d > 320", ctx);
  SqParams p;
  p.qkv = (const char *)qkv; p.q = (const char *)q; p.out = out;
  p.n_out = n_out; p.heads = heads; p.W = heads * 64;
  p.L = L; p.s2 = s2; p.a0 = a0; p.a1 = a1; p.a2 = a2; p.a3 = a3; p.pstride = pstride;
  p.eot = eot; p.offs = offs; p.ctx = ctx;
  ProfScope prof(VTC_PROF_ATTN, 4.0 * (eot ? 0.5 * ctx : L) * 64 * (double)n_out * heads, stream);
  const dim3 grid(cdiv(n_out * heads, 4));
  if (dtype == VTC_F16) hipLaunchKernelGGL((sq_attn_kernel<f16_t>), grid, dim3(256), 0, stream, p);
  else if (dtype == VTC_BF16) hipLaunchKernelGGL((sq_attn_kernel<bf16_t>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((sq_attn_kernel<float>), grid, dim3(256), 0, stream, p);
  VTC_LAUNCH_CHECK("single_query_attention");
  return 0;
}

int launch_mean_cast(const float *x, void *out, int n, int F, int W, int dtype, hipStream_t stream) {
  VTC_CHECK(n > 0 && F > 0 && W > 0, "mean_cast: n=%d F=%d W=%d", n, F, W);
  const dim3 grid(cdiv(n * W, 256));
  if (dtype == VTC_F16) hipLaunchKernelGGL((mean_cast_kernel<f16_t>), grid, dim3(256), 0, stream, x, (f16_t *)out, n, F, W);
  else if (dtype == VTC_BF16) hipLaunchKernelGGL((mean_cast_kernel<bf16_t>), grid, dim3(256), 0, stream, x, (bf16_t *)out, n, F, W);
  else hipLaunchKernelGGL((mean_cast_kernel<float>), grid, dim3(256), 0, stream, x, (float *)out, n, F, W);
  VTC_LAUNCH_CHECK("mean_cast");
  return 0;
}

int launch_cls_global_attention(const void *qkv, void *out, int n_items, int Ttok, int heads, int dtype, hipStream_t stream) {
  VTC_CHECK(Ttok <= 1024, "cls attention: %d tokens > 1024", Ttok);
  VTC_CHECK(dtype == VTC_BF16 || dtype == VTC_F32 || dtype == VTC_F16, "cls attention: dtype %d", dtype);
  const int total = n_items * heads;
  ProfScope prof(VTC_PROF_ATTN, 4.0 * Ttok * 64 * (double)total, stream);
  if (dtype == VTC_F16)
    hipLaunchKernelGGL((cls_global_attn_kernel<f16_t>), dim3(cdiv(total, 4)), dim3(256), 4 * 1024 * sizeof(float), stream,
                       (const f16_t *)qkv, (f16_t *)out, n_items, Ttok, heads);
  else if (dtype == VTC_BF16)
    hipLaunchKernelGGL((cls_global_attn_kernel<bf16_t>), dim3(cdiv(total, 4)), dim3(256), 4 * 1024 * sizeof(float), stream,
                       (const bf16_t *)qkv, (bf16_t *)out, n_items, Ttok, heads);
  else
    hipLaunchKernelGGL((cls_global_attn_kernel<float>), dim3(cdiv(total, 4)), dim3(256), 4 * 1024 * sizeof(float), stream,
                       (const float *)qkv, (float *)out, n_items, Ttok, heads);
  VTC_LAUNCH_CHECK("cls_global_attention");
  return 0;
}

extern "C" int vtc_attention(const void *qkv, void *out, float *cls_out, int n_seq, int L, int heads, int causal, int s2,
                             int a0, int a1, int a2, int a3, int pstride, int dtype, void *stream) {
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "attention: bad dtype %d", dtype);
  return launch_attention(qkv, out, cls_out, n_seq, L, heads, causal, s2, a0, a1, a2, a3, pstride, dtype,
                          (hipStream_t)stream);
}

extern "C" int vtc_single_query_attention(const void *qkv, const void *q, float *out, int n_out, int L, int heads, int s2, int a0, int a1,
                                          int a2, int a3, int pstride, const int *eot, const int *offs, int ctx, int dtype, void *stream) {
  VTC_CHECK(qkv && q && out, "single_query_attention: null argument");
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "single_query_attention: bad dtype %d", dtype);
  return launch_single_query_attention(qkv, q, out, n_out, L, heads, s2, a0, a1, a2, a3, pstride, eot, offs, ctx, dtype, (hipStream_t)stream);
}
