// norm.hip -- row-wise fp32 kernels (x and y may alias: ln_pre runs in place)
// row-wise fp32 kernels: LayerNorm (+ row gather, + cast-only), L2 normalise,
// group mean.  One 64-lane wave per row, float4 accesses, shuffle reductions.
// Replaces LayerNorm (model/timesformer_clip_alt.py:22-28, upstream LayerNorm),
// normalize (model/model.py:26-27) and the frame / title+comment means (:338, :357-362).
#include <algorithm>

#include "common.h"
#include "ln_row.h"

namespace {

// One wave per row; a lane owns chunks of EIGHT consecutive columns (two 16-byte loads, and for bf16 output one
// 16-byte store: 8-byte stores run at 0.54-0.70 of the 16-byte rate on this memory system).  The arithmetic lives in
// ln_row.h: the residual GEMM's fused LayerNorm (gemm.hip, EPI_RESID_LN) must produce the same bits.
template <typename OutT, bool NO_NORM>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, OutT *y, int rows,
                                                        int width, const int *__restrict__ row_index, int row_mul,
                                                        const int *__restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (r >= rows) return;
  const size_t src = row_index ? (size_t)row_index[r] : (size_t)r * row_mul;
  const float *xr = x + src * width;
  float4 v[LN_MAXV][2];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      v[i][0] = *reinterpret_cast<const float4 *>(xr + c);
      v[i][1] = *reinterpret_cast<const float4 *>(xr + c + 4);
    }
  }
  ln_row_compute<OutT, NO_NORM>(v, gamma, beta, y + (size_t)r * width, width, lane);
}

// ---- folded LayerNorm (GemmEpi::fold_*): the row statistics a consuming GEMM's epilogue applies -----------------------------
// stat[r] = (mean, rstd) of row r from the residual GEMM's per-(64 columns, row) partials (sum, squared deviations from the
// partial mean), merged as a two-pass variance would be: M2 = sum_p M2_p + 64 (mean_p - mean)^2.  One thread per row.
__global__ __launch_bounds__(256) void fold_stats_kernel(const float2 *__restrict__ part, int nb, int rows, float2 *__restrict__ stat,
                                                         const int *__restrict__ rows_dev) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (rows_dev) rows = *rows_dev;       // also the stride between the partial planes (the producing GEMM's M)
  if (r >= rows) return;
  float s = 0.f;
  for (int p = 0; p < nb; ++p) s += part[(size_t)p * rows + r].x;
  const float mean = s / (64.0f * nb);
  float m2 = 0.f;
  for (int p = 0; p < nb; ++p) {
    const float2 v = part[(size_t)p * rows + r];
    const float d = v.x * (1.0f / 64.0f) - mean;
    m2 += v.y + 64.0f * d * d;
  }
  stat[r] = make_float2(mean, 1.0f / sqrtf(m2 / (64.0f * nb) + 1e-5f));
}

// Layer 0 has no residual GEMM in front of it: y16 = x in the operand format, stat = (mean, rstd) as LayerNorm computes them.
template <typename OutT>
__global__ __launch_bounds__(256) void cast_rowstats_kernel(const float *__restrict__ x, OutT *__restrict__ y, OutT *__restrict__ ylo,
                                                            float2 *__restrict__ stat, int rows, int width,
                                                            const int *__restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (r >= rows) return;
  const float *xr = x + (size_t)r * width;
  float4 v[LN_MAXV][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      v[i][0] = *reinterpret_cast<const float4 *>(xr + c);
      v[i][1] = *reinterpret_cast<const float4 *>(xr + c + 4);
      s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
  }
  const float mean = wave_sum(s) / width;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float a = v[i][h].x - mean, b = v[i][h].y - mean, cc = v[i][h].z - mean, d = v[i][h].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
      // the stream is stored CENTRED (x - mean): its only readers are LayerNorms (gemm.hip, SPLIT)
      const float o[8] = {v[i][0].x - mean, v[i][0].y - mean, v[i][0].z - mean, v[i][0].w - mean,
                          v[i][1].x - mean, v[i][1].y - mean, v[i][1].z - mean, v[i][1].w - mean};
      uint4 pk;
      pk.x = (unsigned)cvt16<OutT>(o[0]) | ((unsigned)cvt16<OutT>(o[1]) << 16);
      pk.y = (unsigned)cvt16<OutT>(o[2]) | ((unsigned)cvt16<OutT>(o[3]) << 16);
      pk.z = (unsigned)cvt16<OutT>(o[4]) | ((unsigned)cvt16<OutT>(o[5]) << 16);
      pk.w = (unsigned)cvt16<OutT>(o[6]) | ((unsigned)cvt16<OutT>(o[7]) << 16);
      *reinterpret_cast<uint4 *>(y + (size_t)r * width + c) = pk;
      // lo = fmt(x - hi): the pair carries the row to 2^-17 (bf16) / 2^-22 (half) relative
      const unsigned hw[4] = {pk.x, pk.y, pk.z, pk.w};
      unsigned lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float l0 = o[2 * e] - up16<OutT>((unsigned short)(hw[e] & 0xFFFFu)), l1 = o[2 * e + 1] - up16<OutT>((unsigned short)(hw[e] >> 16));
        lw[e] = (unsigned)cvt16<OutT>(l0) | ((unsigned)cvt16<OutT>(l1) << 16);
      }
      *reinterpret_cast<uint4 *>(ylo + (size_t)r * width + c) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / width + 1e-5f);
  if (lane == 0) stat[r] = make_float2(0.0f, rstd);       // the mean of the stored (centred) row
}

// ln_pre + the entry into the folded scheme in ONE pass over the rows (round 6): z = LayerNorm(x; gamma, beta) -- the arithmetic of
// ln_row.h -- and then what cast_rowstats_kernel does with z (centred (hi, lo) pair + rstd), without the fp32 z ever reaching HBM:
// at 1 024 videos the two launches moved 2 x 2.5 GB, this one reads 1.24 GB and writes 1.24 GB.
template <typename OutT>
__global__ __launch_bounds__(256) void ln_cast_rowstats_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                               OutT *__restrict__ y, OutT *__restrict__ ylo, float2 *__restrict__ stat, int rows, int width) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float *xr = x + (size_t)r * width;
  float4 v[LN_MAXV][2];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      v[i][0] = *reinterpret_cast<const float4 *>(xr + c);
      v[i][1] = *reinterpret_cast<const float4 *>(xr + c + 4);
      s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
  }
  const float mean = wave_sum(s) / width;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if ((lane + 64 * i) * 8 < width) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float a = v[i][h].x - mean, b = v[i][h].y - mean, cc = v[i][h].z - mean, d = v[i][h].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / width + 1e-5f);
  // z = LayerNorm(x) in place of v, and its row sum
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      const float4 g0 = *reinterpret_cast<const float4 *>(gamma + c), g1 = *reinterpret_cast<const float4 *>(gamma + c + 4);
      const float4 b0 = *reinterpret_cast<const float4 *>(beta + c), b1 = *reinterpret_cast<const float4 *>(beta + c + 4);
      v[i][0] = make_float4((v[i][0].x - mean) * rstd * g0.x + b0.x, (v[i][0].y - mean) * rstd * g0.y + b0.y,
                            (v[i][0].z - mean) * rstd * g0.z + b0.z, (v[i][0].w - mean) * rstd * g0.w + b0.w);
      v[i][1] = make_float4((v[i][1].x - mean) * rstd * g1.x + b1.x, (v[i][1].y - mean) * rstd * g1.y + b1.y,
                            (v[i][1].z - mean) * rstd * g1.z + b1.z, (v[i][1].w - mean) * rstd * g1.w + b1.w);
      s2 += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
    }
  }
  const float mean2 = wave_sum(s2) / width;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      const float o[8] = {v[i][0].x - mean2, v[i][0].y - mean2, v[i][0].z - mean2, v[i][0].w - mean2,
                          v[i][1].x - mean2, v[i][1].y - mean2, v[i][1].z - mean2, v[i][1].w - mean2};
#pragma unroll
      for (int e = 0; e < 8; ++e) q2 += o[e] * o[e];
      uint4 pk;
      pk.x = (unsigned)cvt16<OutT>(o[0]) | ((unsigned)cvt16<OutT>(o[1]) << 16);
      pk.y = (unsigned)cvt16<OutT>(o[2]) | ((unsigned)cvt16<OutT>(o[3]) << 16);
      pk.z = (unsigned)cvt16<OutT>(o[4]) | ((unsigned)cvt16<OutT>(o[5]) << 16);
      pk.w = (unsigned)cvt16<OutT>(o[6]) | ((unsigned)cvt16<OutT>(o[7]) << 16);
      *reinterpret_cast<uint4 *>(y + (size_t)r * width + c) = pk;
      const unsigned hw[4] = {pk.x, pk.y, pk.z, pk.w};
      unsigned lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float l0 = o[2 * e] - up16<OutT>((unsigned short)(hw[e] & 0xFFFFu)), l1 = o[2 * e + 1] - up16<OutT>((unsigned short)(hw[e] >> 16));
        lw[e] = (unsigned)cvt16<OutT>(l0) | ((unsigned)cvt16<OutT>(l1) << 16);
      }
      *reinterpret_cast<uint4 *>(ylo + (size_t)r * width + c) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
  const float rstd2 = 1.0f / sqrtf(wave_sum(q2) / width + 1e-5f);
  if (lane == 0) stat[r] = make_float2(0.0f, rstd2);       // the mean of the stored (centred) row
}

// x[src] = hi[src] + lo[src] for the rows src = row_index[i] (or i * row_mul): the fp32 rows the final LayerNorm reads
template <typename T>
__global__ __launch_bounds__(256) void split_merge_rows_kernel(const T *__restrict__ hi, const T *__restrict__ lo, float *__restrict__ x,
                                                               int n, int width, const int *__restrict__ row_index, int row_mul,
                                                               const int *__restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) n = *rows_dev;
  if (i >= n) return;
  const size_t src = row_index ? (size_t)row_index[i] : (size_t)i * row_mul;
  for (int c = lane * 8; c < width; c += 512) {
    const uint4 h = *reinterpret_cast<const uint4 *>(hi + src * width + c), l = *reinterpret_cast<const uint4 *>(lo + src * width + c);
    const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = up16<T>((unsigned short)(hw[e] & 0xFFFFu)) + up16<T>((unsigned short)(lw[e] & 0xFFFFu));
      o[2 * e + 1] = up16<T>((unsigned short)(hw[e] >> 16)) + up16<T>((unsigned short)(lw[e] >> 16));
    }
    *reinterpret_cast<float4 *>(x + src * width + c) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4 *>(x + src * width + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// ---- split-K finish of a residual GEMM on the folded (hi, lo) stream (small batches; towers.hip resid_proj, gemm.hip GemmEpi::ksplit) ----
// The GEMM left `nsl` planes of partial products part[s][row][c] (fp32).  Per row, one wave: v = bias + sum_s part[s] (slices in order),
// then exactly what the re-centring residual epilogue of gemm.hip does (EPI_RESID_FOLD_C): x = ((hi - mean_prev) + lo) + v, stored back as
// hi = fmt(x), lo = fmt(x - hi) -- and the row's LayerNorm statistics (mean, rstd) of x written WHOLE (two passes over the wave's
// registers), which is what fold_stats_kernel derives from the per-64-column partials: that launch does not exist on this path.
template <typename T>
__global__ __launch_bounds__(256) void splitk_resid_rows_kernel(const float *__restrict__ part, int nsl, int stride, const float *__restrict__ bias,
                                                                T *__restrict__ hi, T *__restrict__ lo, float2 *__restrict__ stat, int rows, int width,
                                                                const int *__restrict__ rows_dev) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rows_dev) rows = *rows_dev;
  if (r >= rows) return;
  const float mu = stat[r].x;                     // the row's mean before this update (the stream is kept centred)
  float4 y[4];                                    // width <= 1024: four columns per lane and 256-column chunk
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < width) {
      float4 v = bias ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sl = 0; sl < nsl; ++sl) {
        const float4 q = *reinterpret_cast<const float4 *>(part + (size_t)sl * stride + (size_t)r * width + c);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      const uint2 h = *reinterpret_cast<const uint2 *>(hi + (size_t)r * width + c), l = *reinterpret_cast<const uint2 *>(lo + (size_t)r * width + c);
      float4 x;
      x.x = ((up16<T>((unsigned short)(h.x & 0xFFFFu)) - mu) + up16<T>((unsigned short)(l.x & 0xFFFFu))) + v.x;
      x.y = ((up16<T>((unsigned short)(h.x >> 16)) - mu) + up16<T>((unsigned short)(l.x >> 16))) + v.y;
      x.z = ((up16<T>((unsigned short)(h.y & 0xFFFFu)) - mu) + up16<T>((unsigned short)(l.y & 0xFFFFu))) + v.z;
      x.w = ((up16<T>((unsigned short)(h.y >> 16)) - mu) + up16<T>((unsigned short)(l.y >> 16))) + v.w;
      uint2 pk, pl;
      pk.x = pack16<T>(x.x, x.y);
      pk.y = pack16<T>(x.z, x.w);
      pl.x = pack16<T>(x.x - up16<T>((unsigned short)(pk.x & 0xFFFFu)), x.y - up16<T>((unsigned short)(pk.x >> 16)));
      pl.y = pack16<T>(x.z - up16<T>((unsigned short)(pk.y & 0xFFFFu)), x.w - up16<T>((unsigned short)(pk.y >> 16)));
      *reinterpret_cast<uint2 *>(hi + (size_t)r * width + c) = pk;
      *reinterpret_cast<uint2 *>(lo + (size_t)r * width + c) = pl;
      y[i] = x;
      s1 += (x.x + x.y) + (x.z + x.w);
    }
  }
  const float mean = wave_sum(s1) / width;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (lane * 4 + 256 * i < width) {
      const float d0 = y[i].x - mean, d1 = y[i].y - mean, d2 = y[i].z - mean, d3 = y[i].w - mean;
      m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  m2 = wave_sum(m2);
  if (lane == 0) stat[r] = make_float2(mean, 1.0f / sqrtf(m2 / width + 1e-5f));
}

// dst[i] = src row (row_index[i] or i * row_mul): the compact copy of the rows that reach a tower's output (towers.hip, last block)
__global__ __launch_bounds__(256) void gather_rows_kernel(const char *__restrict__ src, char *__restrict__ dst, int n, int row_bytes,
                                                          const int *__restrict__ row_index, int row_mul) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const size_t r = row_index ? (size_t)row_index[i] : (size_t)i * row_mul;
  for (int c = lane * 16; c < row_bytes; c += 1024)
    *reinterpret_cast<uint4 *>(dst + (size_t)i * row_bytes + c) = *reinterpret_cast<const uint4 *>(src + r * row_bytes + c);
}

__global__ __launch_bounds__(256) void normalize_kernel(const float *__restrict__ x, float *__restrict__ out, int n, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float *xr = x + (size_t)r * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c] * xr[c];
  const float nrm = sqrtf(wave_sum(s));
  for (int c = lane; c < d; c += 64) out[(size_t)r * d + c] = xr[c] / nrm;   // x / x.norm(): division, as the reference
}

// The last launch of a wrapper's forward: both embedding sets normalised in ONE launch (rows 0 .. nx - 1 from x, the rest from y), and the
// non-finite watchdog with it -- a NaN / inf anywhere in a row makes its squared norm non-finite: flag |= 1 (x) / 2 (y).  Per row the
// arithmetic of normalize_kernel.
__global__ __launch_bounds__(256) void normalize2_kernel(const float *__restrict__ x, float *__restrict__ outx, int nx, const float *__restrict__ y,
                                                         float *__restrict__ outy, int ny, int d, int *flag) {
  const int lane = threadIdx.x & 63;
  int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nx + ny) return;
  const bool second = r >= nx;
  if (second) r -= nx;
  const float *xr = (second ? y : x) + (size_t)r * d;
  float *o = (second ? outy : outx) + (size_t)r * d;
  float s = 0.f;
  for (int c = lane; c < d; c += 64) s += xr[c] * xr[c];
  s = wave_sum(s);
  const float nrm = sqrtf(s);
  for (int c = lane; c < d; c += 64) o[c] = xr[c] / nrm;   // x / x.norm(): division, as the reference
  if (flag && lane == 0 && (__float_as_uint(s) & 0x7F800000u) == 0x7F800000u) atomicOr(flag, second ? 2 : 1);
}

__global__ __launch_bounds__(256) void mean_groups_kernel(const float *__restrict__ x, float *__restrict__ out, int n_groups,
                                                          int group, int d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_groups * d) return;
  const int gi = i / d, c = i - gi * d;
  float s = 0.f;
  for (int k = 0; k < group; ++k) s += x[((size_t)gi * group + k) * d + c];
  out[i] = s / group;
}

// out[g] = (a[g] + sum_k b[g * group + k]) / (1 + group): title + its comments, summed in the reference's order
// (torch.mean(torch.cat([title[None], comments]), 0), model/model.py:357-362)
__global__ __launch_bounds__(256) void mean_head_groups_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               float *__restrict__ out, int n_groups, int group, int d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_groups * d) return;
  const int gi = i / d, c = i - gi * d;
  float s = a[i];
  for (int k = 0; k < group; ++k) s += b[((size_t)gi * group + k) * d + c];
  out[i] = s / (1 + group);
}

// out[i] = mean of rows [offsets[i], offsets[i+1]) (ragged groups: chunks of one video)
__global__ __launch_bounds__(256) void segment_mean_kernel(const float *__restrict__ x, const int *__restrict__ offsets,
                                                           float *__restrict__ out, int n, int d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * d) return;
  const int gi = i / d, c = i - gi * d;
  const int lo = offsets[gi], hi = offsets[gi + 1];
  float s = 0.f;
  for (int k = lo; k < hi; ++k) s += x[(size_t)k * d + c];
  out[i] = s / (float)(hi - lo);
}

}  // namespace

extern "C" int vtc_segment_mean(const float *x, const int *offsets, float *out, int n_groups, int d, void *stream) {
  VTC_CHECK(n_groups > 0 && d > 0, "segment_mean: bad sizes");
  hipLaunchKernelGGL(segment_mean_kernel, dim3(cdiv(n_groups * d, 256)), dim3(256), 0, (hipStream_t)stream, x, offsets, out,
                     n_groups, d);
  VTC_LAUNCH_CHECK("segment_mean");
  return 0;
}

int launch_layernorm(const float *x, const float *g, const float *b, void *y, int rows, int width, int out_dtype,
                     const int *row_index, int row_mul, bool no_norm, hipStream_t stream, const int *rows_dev) {
  VTC_CHECK(rows > 0, "layernorm: rows=%d", rows);
  VTC_CHECK(width % 8 == 0 && width <= 512 * LN_MAXV, "layernorm: width=%d unsupported (multiple of 8, <= 1024)", width);
  const dim3 grid(cdiv(rows, 4)), block(256);
  ProfScope prof(VTC_PROF_NORM, (double)rows * width * (4 + (out_dtype != VTC_F32 ? 2 : 4)), stream);
  if (out_dtype == VTC_F16) {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<f16_t, true>), grid, block, 0, stream, x, g, b, (f16_t *)y, rows, width, row_index, row_mul, rows_dev);
    else hipLaunchKernelGGL((layernorm_kernel<f16_t, false>), grid, block, 0, stream, x, g, b, (f16_t *)y, rows, width, row_index, row_mul, rows_dev);
  } else if (out_dtype == VTC_BF16) {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<bf16_t, true>), grid, block, 0, stream, x, g, b, (bf16_t *)y, rows, width, row_index, row_mul, rows_dev);
    else hipLaunchKernelGGL((layernorm_kernel<bf16_t, false>), grid, block, 0, stream, x, g, b, (bf16_t *)y, rows, width, row_index, row_mul, rows_dev);
  } else {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<float, true>), grid, block, 0, stream, x, g, b, (float *)y, rows, width, row_index, row_mul, rows_dev);
    else hipLaunchKernelGGL((layernorm_kernel<float, false>), grid, block, 0, stream, x, g, b, (float *)y, rows, width, row_index, row_mul, rows_dev);
  }
  VTC_LAUNCH_CHECK("layernorm");
  return 0;
}

int launch_fold_stats(const float *part, int nb, int rows, float *stat, hipStream_t stream, const int *rows_dev) {
  ProfScope prof(VTC_PROF_NORM, (double)rows * 8 * (nb + 1), stream);
  hipLaunchKernelGGL(fold_stats_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, stream, (const float2 *)part, nb, rows, (float2 *)stat, rows_dev);
  VTC_LAUNCH_CHECK("fold_stats");
  return 0;
}

int launch_splitk_resid_rows(const float *part, int nsl, int stride, const float *bias, void *hi, void *lo, float *stat, int rows, int width, int dtype,
                             hipStream_t stream, const int *rows_dev) {
  VTC_CHECK(width % 256 == 0 && width <= 1024 && nsl >= 2 && (dtype == VTC_BF16 || dtype == VTC_F16), "splitk_resid_rows: width=%d slices=%d dtype=%d", width, nsl, dtype);
  ProfScope prof(VTC_PROF_NORM, (double)rows * width * (4.0 * nsl + 8.0), stream);
  if (dtype == VTC_F16)
    hipLaunchKernelGGL((splitk_resid_rows_kernel<f16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, part, nsl, stride, bias, (f16_t *)hi, (f16_t *)lo, (float2 *)stat, rows, width, rows_dev);
  else
    hipLaunchKernelGGL((splitk_resid_rows_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, part, nsl, stride, bias, (bf16_t *)hi, (bf16_t *)lo, (float2 *)stat, rows, width, rows_dev);
  VTC_LAUNCH_CHECK("splitk_resid_rows");
  return 0;
}

int launch_cast_rowstats(const float *x, void *y16, void *y16lo, float *stat, int rows, int width, int dtype, hipStream_t stream,
                         const int *rows_dev) {
  VTC_CHECK(width % 8 == 0 && width <= 512 * LN_MAXV && (dtype == VTC_BF16 || dtype == VTC_F16), "cast_rowstats: width=%d dtype=%d", width, dtype);
  ProfScope prof(VTC_PROF_NORM, (double)rows * width * 8, stream);
  if (dtype == VTC_F16)
    hipLaunchKernelGGL((cast_rowstats_kernel<f16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, (f16_t *)y16, (f16_t *)y16lo, (float2 *)stat, rows, width, rows_dev);
  else
    hipLaunchKernelGGL((cast_rowstats_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, (bf16_t *)y16, (bf16_t *)y16lo, (float2 *)stat, rows, width, rows_dev);
  VTC_LAUNCH_CHECK("cast_rowstats");
  return 0;
}

int launch_ln_cast_rowstats(const float *x, const float *gamma, const float *beta, void *y16, void *y16lo, float *stat, int rows, int width, int dtype,
                            hipStream_t stream) {
  VTC_CHECK(width % 8 == 0 && width <= 512 * LN_MAXV && (dtype == VTC_BF16 || dtype == VTC_F16), "ln_cast_rowstats: width=%d dtype=%d", width, dtype);
  ProfScope prof(VTC_PROF_NORM, (double)rows * width * 8, stream);
  if (dtype == VTC_F16)
    hipLaunchKernelGGL((ln_cast_rowstats_kernel<f16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, gamma, beta, (f16_t *)y16, (f16_t *)y16lo, (float2 *)stat, rows, width);
  else
    hipLaunchKernelGGL((ln_cast_rowstats_kernel<bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, gamma, beta, (bf16_t *)y16, (bf16_t *)y16lo, (float2 *)stat, rows, width);
  VTC_LAUNCH_CHECK("ln_cast_rowstats");
  return 0;
}

int launch_gather_rows(const void *src, void *dst, int n, int row_bytes, const int *row_index, int row_mul, hipStream_t stream) {
  VTC_CHECK(n > 0 && row_bytes > 0 && row_bytes % 16 == 0, "gather_rows: n=%d row_bytes=%d", n, row_bytes);
  ProfScope prof(VTC_PROF_NORM, (double)n * row_bytes * 2, stream);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n, 4)), dim3(256), 0, stream, (const char *)src, (char *)dst, n, row_bytes, row_index, row_mul);
  VTC_LAUNCH_CHECK("gather_rows");
  return 0;
}

int launch_split_merge_rows(const void *hi, const void *lo, float *x, int n, int width, const int *row_index, int row_mul, int dtype,
                            hipStream_t stream, const int *rows_dev) {
  VTC_CHECK(width % 8 == 0 && (dtype == VTC_BF16 || dtype == VTC_F16), "split_merge_rows: width=%d dtype=%d", width, dtype);
  ProfScope prof(VTC_PROF_NORM, (double)n * width * 8, stream);
  if (dtype == VTC_F16)
    hipLaunchKernelGGL((split_merge_rows_kernel<f16_t>), dim3(cdiv(n, 4)), dim3(256), 0, stream, (const f16_t *)hi, (const f16_t *)lo, x, n, width, row_index, row_mul, rows_dev);
  else
    hipLaunchKernelGGL((split_merge_rows_kernel<bf16_t>), dim3(cdiv(n, 4)), dim3(256), 0, stream, (const bf16_t *)hi, (const bf16_t *)lo, x, n, width, row_index, row_mul, rows_dev);
  VTC_LAUNCH_CHECK("split_merge_rows");
  return 0;
}

extern "C" int vtc_layernorm(const float *x, const float *g, const float *b, void *y, int rows, int width, int out_dtype,
                             const int *row_index, int row_mul, void *stream) {
  return launch_layernorm(x, g, b, y, rows, width, out_dtype, row_index, row_mul, false, (hipStream_t)stream);
}

extern "C" int vtc_normalize_rows(const float *x, float *out, int n, int d, void *stream) {
  VTC_CHECK(n > 0 && d > 0, "normalize_rows: n=%d d=%d", n, d);
  hipLaunchKernelGGL(normalize_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, out, n, d);
  VTC_LAUNCH_CHECK("normalize_rows");
  return 0;
}

extern "C" int vtc_normalize_rows2(const float *x, float *outx, int nx, const float *y, float *outy, int ny, int d, int *flag, void *stream) {
  VTC_CHECK(x && outx && y && outy && nx > 0 && ny > 0 && d > 0, "normalize_rows2: bad arguments (nx=%d ny=%d d=%d)", nx, ny, d);
  hipLaunchKernelGGL(normalize2_kernel, dim3(cdiv(nx + ny, 4)), dim3(256), 0, (hipStream_t)stream, x, outx, nx, y, outy, ny, d, flag);
  VTC_LAUNCH_CHECK("normalize_rows2");
  return 0;
}

// flag[0] |= 1 when x holds a non-finite value.  `flag` may live in pinned host memory (the host then reads it without a sync).
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float *__restrict__ x, size_t n, int *flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(x[i]);
    bad |= (u & 0x7F800000u) == 0x7F800000u;
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int vtc_nonfinite_flag(const float *x, size_t n, int *flag, void *stream) {
  VTC_CHECK(x && flag && n > 0, "nonfinite_flag: bad arguments");
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, flag);
  VTC_LAUNCH_CHECK("nonfinite_flag");
  return 0;
}

// flag[0] |= 1 when x holds a non-finite value, |= 2 when y does: the wrappers' watchdog over the two embedding sets a forward returns (one launch)
__global__ __launch_bounds__(256) void nonfinite_flag2_kernel(const float *__restrict__ x, size_t n, const float *__restrict__ y, size_t m, int *flag) {
  bool bx = false, by = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n + m; i += (size_t)gridDim.x * 256) {
    const unsigned u = __float_as_uint(i < n ? x[i] : y[i - n]);
    const bool bad = (u & 0x7F800000u) == 0x7F800000u;
    bx |= bad && i < n;
    by |= bad && i >= n;
  }
  const int bits = (__ballot(bx) != 0 ? 1 : 0) | (__ballot(by) != 0 ? 2 : 0);
  if (bits && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
}

extern "C" int vtc_nonfinite_flag2(const float *x, size_t n, const float *y, size_t m, int *flag, void *stream) {
  VTC_CHECK(x && y && flag && n > 0 && m > 0, "nonfinite_flag2: bad arguments");
  const unsigned grid = (unsigned)std::min<size_t>((n + m + 255) / 256, 1024);
  hipLaunchKernelGGL(nonfinite_flag2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, y, m, flag);
  VTC_LAUNCH_CHECK("nonfinite_flag2");
  return 0;
}

extern "C" int vtc_mean_head_groups(const float *a, const float *b, float *out, int n_groups, int group, int d, void *stream) {
  VTC_CHECK(n_groups > 0 && group >= 0 && d > 0 && a && out && (b || group == 0), "mean_head_groups: bad arguments");
  hipLaunchKernelGGL(mean_head_groups_kernel, dim3(cdiv(n_groups * d, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n_groups, group, d);
  VTC_LAUNCH_CHECK("mean_head_groups");
  return 0;
}

extern "C" int vtc_mean_groups(const float *x, float *out, int n_groups, int group, int d, void *stream) {
  VTC_CHECK(n_groups > 0 && group > 0 && d > 0, "mean_groups: bad sizes");
  hipLaunchKernelGGL(mean_groups_kernel, dim3(cdiv(n_groups * d, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     n_groups, group, d);
  VTC_LAUNCH_CHECK("mean_groups");
  return 0;
}
