// norm.hip -- row-wise fp32 kernels (x and y may alias: ln_pre runs in place)
// row-wise fp32 kernels: LayerNorm (+ row gather, + cast-only), L2 normalise,
// group mean.  One 64-lane wave per row, float4 accesses, shuffle reductions.
// Replaces LayerNorm (model/timesformer_clip_alt.py:22-28, upstream LayerNorm),
// normalize (model/model.py:26-27) and the frame / title+comment means (:338, :357-362).
#include "common.h"
#include "ln_row.h"

namespace {

// One wave per row; a lane owns chunks of EIGHT consecutive columns (two 16-byte loads, and for bf16 output one
// 16-byte store: 8-byte stores run at 0.54-0.70 of the 16-byte rate on this memory system).  The arithmetic lives in
// ln_row.h: the residual GEMM's fused LayerNorm (gemm.hip, EPI_RESID_LN) must produce the same bits.
template <typename OutT, bool NO_NORM>
__global__ __launch_bounds__(256) void layernorm_kernel(const float *x, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, OutT *y, int rows,
                                                        int width, const int *__restrict__ row_index, int row_mul) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
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

__global__ __launch_bounds__(256) void mean_groups_kernel(const float *__restrict__ x, float *__restrict__ out, int n_groups,
                                                          int group, int d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_groups * d) return;
  const int gi = i / d, c = i - gi * d;
  float s = 0.f;
  for (int k = 0; k < group; ++k) s += x[((size_t)gi * group + k) * d + c];
  out[i] = s / group;
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
                     const int *row_index, int row_mul, bool no_norm, hipStream_t stream) {
  VTC_CHECK(rows > 0, "layernorm: rows=%d", rows);
  VTC_CHECK(width % 8 == 0 && width <= 512 * LN_MAXV, "layernorm: width=%d unsupported (multiple of 8, <= 1024)", width);
  const dim3 grid(cdiv(rows, 4)), block(256);
  ProfScope prof(VTC_PROF_NORM, (double)rows * width * (4 + (out_dtype != VTC_F32 ? 2 : 4)), stream);
  if (out_dtype == VTC_F16) {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<f16_t, true>), grid, block, 0, stream, x, g, b, (f16_t *)y, rows, width, row_index, row_mul);
    else hipLaunchKernelGGL((layernorm_kernel<f16_t, false>), grid, block, 0, stream, x, g, b, (f16_t *)y, rows, width, row_index, row_mul);
  } else if (out_dtype == VTC_BF16) {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<bf16_t, true>), grid, block, 0, stream, x, g, b, (bf16_t *)y, rows, width, row_index, row_mul);
    else hipLaunchKernelGGL((layernorm_kernel<bf16_t, false>), grid, block, 0, stream, x, g, b, (bf16_t *)y, rows, width, row_index, row_mul);
  } else {
    if (no_norm) hipLaunchKernelGGL((layernorm_kernel<float, true>), grid, block, 0, stream, x, g, b, (float *)y, rows, width, row_index, row_mul);
    else hipLaunchKernelGGL((layernorm_kernel<float, false>), grid, block, 0, stream, x, g, b, (float *)y, rows, width, row_index, row_mul);
  }
  VTC_LAUNCH_CHECK("layernorm");
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

extern "C" int vtc_mean_groups(const float *x, float *out, int n_groups, int group, int d, void *stream) {
  VTC_CHECK(n_groups > 0 && group > 0 && d > 0, "mean_groups: bad sizes");
  hipLaunchKernelGGL(mean_groups_kernel, dim3(cdiv(n_groups * d, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     n_groups, group, d);
  VTC_LAUNCH_CHECK("mean_groups");
  return 0;
}
