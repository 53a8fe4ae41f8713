// train.hip -- backward and optimizer kernels of the adapter-only training step:
// PretrainedCLIP_finaltf with the towers frozen (configs/pretrained_clip_comments_attn_frozen.jsonc), clip_loss
// (model/loss.py:18-22), torch.optim.Adam(amsgrad=True).  The trainable set is the 2-layer Context Adapter Module
// (model/model.py:396-400) -- 6 tokens per item, width 512 -- so every kernel here is small and fp32; the matrix
// products of the backward pass (dgrad, wgrad) reuse vtc_gemm on transposed operands (vtc_transpose_f32).
// Host orchestration: vtc_amd/host/adapter_train.py; oracle: oracle/train_ref.py.
#include "common.h"

#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ x, float *__restrict__ y, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? x[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < cols && r < rows) y[(size_t)c * rows + r] = tile[tx][j];
  }
}

// out[c] (+)= sum_r x[r][c]; one thread per column, rows strided over gridDim.y with an atomic merge
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, float *__restrict__ out, int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) s += x[(size_t)r * cols + c];
  atomicAdd(out + c, s);
}

// LayerNorm backward (eps 1e-5, biased variance): one wave per row, width <= 1024.
//   xhat = (x - mean) rstd;  g = dy * gamma;  dx = rstd (g - mean(g) - xhat mean(g xhat))
//   dgamma += dy * xhat;  dbeta += dy   (atomics over rows)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                            const float *__restrict__ dy, float *__restrict__ dx,
                                                            float *__restrict__ dgamma, float *__restrict__ dbeta, int rows, int width,
                                                            int accumulate_dx) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float *xr = x + (size_t)r * width, *dyr = dy + (size_t)r * width;
  float xv[16], gv[16], dv[16];
  float s = 0.f;
  int cnt = 0;
  for (int c = lane; c < width; c += 64, ++cnt) { xv[cnt] = xr[c]; dv[cnt] = dyr[c]; s += xv[cnt]; }
  const float mean = wave_sum(s) / width;
  float q = 0.f;
  for (int i = 0; i < cnt; ++i) { const float a = xv[i] - mean; q += a * a; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / width + 1e-5f);
  float sg = 0.f, sgx = 0.f;
  for (int i = 0; i < cnt; ++i) {
    const int c = lane + 64 * i;
    const float xh = (xv[i] - mean) * rstd;
    gv[i] = dv[i] * gamma[c];
    sg += gv[i];
    sgx += gv[i] * xh;
    xv[i] = xh;
  }
  sg = wave_sum(sg) / width;
  sgx = wave_sum(sgx) / width;
  for (int i = 0; i < cnt; ++i) {
    const int c = lane + 64 * i;
    const float v = rstd * (gv[i] - sg - xv[i] * sgx);
    float *o = dx + (size_t)r * width + c;
    *o = accumulate_dx ? *o + v : v;
    atomicAdd(dgamma + c, dv[i] * xv[i]);
    atomicAdd(dbeta + c, dv[i]);
  }
}

// Unmasked attention backward for short sequences (L <= 16), head_dim 64, item-major contiguous rows:
// one wave per (sequence, head); lane = head dimension.  P = softmax(q k^T / 8), o = P v.
//   dv = P^T do;  dP = do v^T;  dS = P (dP - rowsum(dP P));  dq = dS k / 8;  dk = dS^T q / 8
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                             float *__restrict__ dqkv, int n_seq, int L, int heads) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_seq * heads) return;
  const int s = w / heads, h = w - s * heads;
  const int W = heads * 64;
  float q[16], k[16], v[16], dO[16];
  for (int t = 0; t < L; ++t) {
    const size_t row = (size_t)s * L + t;
    q[t] = qkv[row * 3 * W + h * 64 + lane];
    k[t] = qkv[row * 3 * W + W + h * 64 + lane];
    v[t] = qkv[row * 3 * W + 2 * W + h * 64 + lane];
    dO[t] = dout[row * W + h * 64 + lane];
  }
  float dq[16], dk[16], dv[16];
  for (int t = 0; t < L; ++t) { dq[t] = 0.f; dk[t] = 0.f; dv[t] = 0.f; }
  for (int i = 0; i < L; ++i) {                 // query i
    float p[16], dp[16];
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) { p[j] = wave_sum(q[i] * k[j]) * 0.125f; mx = fmaxf(mx, p[j]); }
    float sum = 0.f;
    for (int j = 0; j < L; ++j) { p[j] = expf(p[j] - mx); sum += p[j]; }
    float dot = 0.f;
    for (int j = 0; j < L; ++j) {
      p[j] /= sum;
      dp[j] = wave_sum(dO[i] * v[j]);
      dot += dp[j] * p[j];
    }
    for (int j = 0; j < L; ++j) {
      const float ds = p[j] * (dp[j] - dot) * 0.125f;
      dv[j] += p[j] * dO[i];
      dq[i] += ds * k[j];
      dk[j] += ds * q[i];
    }
  }
  for (int t = 0; t < L; ++t) {
    const size_t row = (size_t)s * L + t;
    dqkv[row * 3 * W + h * 64 + lane] = dq[t];
    dqkv[row * 3 * W + W + h * 64 + lane] = dk[t];
    dqkv[row * 3 * W + 2 * W + h * 64 + lane] = dv[t];
  }
}

// QuickGELU y = x sigmoid(1.702 x) (timesformer_clip_alt.py:31-33): forward and dx = dy (s + 1.702 x s (1 - s))
__global__ __launch_bounds__(256) void quickgelu_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ out,
                                                        size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], s = 1.0f / (1.0f + expf(-1.702f * v));
  out[i] = dy ? dy[i] * (s + 1.702f * v * s * (1.0f - s)) : v * s;
}

// y = x / |x| (model/model.py:26-27):  dx = (dy - y (y . dy)) / |x|, one wave per row
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ dx,
                                                            int n, int d) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float *xr = x + (size_t)r * d, *dr = dy + (size_t)r * d;
  float s = 0.f, t = 0.f;
  for (int c = lane; c < d; c += 64) { s += xr[c] * xr[c]; t += xr[c] * dr[c]; }
  const float nrm = sqrtf(wave_sum(s));
  const float ydy = wave_sum(t) / nrm;                       // y . dy
  for (int c = lane; c < d; c += 64) dx[(size_t)r * d + c] = (dr[c] - (xr[c] / nrm) * ydy) / nrm;
}

// clip_loss backward: dsim[i][j] = 0.5/n (softmax_row_i[j] + softmax_col_j[i] - 2 [i == j]).
// stats[0..n) = row max, [n..2n) = row sum-exp, [2n..3n) = col max, [3n..4n) = col sum-exp
__global__ __launch_bounds__(256) void lse_stats_kernel(const float *__restrict__ sim, int n, float *__restrict__ stats) {
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
  if (lane == 0) { stats[(col ? 2 * n : 0) + i] = mx; stats[(col ? 3 * n : n) + i] = s; }
}
__global__ __launch_bounds__(256) void clip_loss_bwd_kernel(const float *__restrict__ sim, int n, const float *__restrict__ stats,
                                                            float *__restrict__ dsim) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
  const float v = sim[idx];
  const float pr = expf(v - stats[i]) / stats[n + i], pc = expf(v - stats[2 * n + j]) / stats[3 * n + j];
  dsim[idx] = (0.5f / n) * (pr + pc - (i == j ? 2.0f : 0.0f));
}

// torch.optim.Adam single-tensor step (weight_decay 0), optional amsgrad
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, float *__restrict__ vmax, size_t n, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt, int amsgrad) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = m[i] = b1 * m[i] + (1.0f - b1) * gi;
  float vi = v[i] = b2 * v[i] + (1.0f - b2) * gi * gi;
  if (amsgrad) vi = vmax[i] = fmaxf(vmax[i], vi);
  p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

// out = a x + b y (y may be null)
__global__ __launch_bounds__(256) void axpby_kernel(float *__restrict__ out, const float *__restrict__ x, const float *__restrict__ y, float a,
                                                    float b, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// x[r][:] *= s[r / group]
__global__ __launch_bounds__(256) void scale_rows_kernel(float *__restrict__ x, const float *__restrict__ s, int rows, int d, int group) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)rows * d) x[i] *= s[(i / d) / group];
}

}  // namespace

#define GRID1(n) dim3((unsigned)(((size_t)(n) + 255) / 256))

extern "C" int vtc_transpose_f32(const float *x, float *y, int rows, int cols, void *stream) {
  VTC_CHECK(x && y && rows > 0 && cols > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
  VTC_LAUNCH_CHECK("transpose");
  return 0;
}

extern "C" int vtc_colsum_f32(const float *x, float *out, int rows, int cols, void *stream) {
  VTC_CHECK(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
  (void)hipMemsetAsync(out, 0, (size_t)cols * 4, (hipStream_t)stream);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 256), std::min(rows, 64)), dim3(256), 0, (hipStream_t)stream, x, out, rows, cols);
  VTC_LAUNCH_CHECK("colsum");
  return 0;
}

extern "C" int vtc_layernorm_bwd(const float *x, const float *gamma, const float *dy, float *dx, float *dgamma, float *dbeta, int rows,
                                 int width, int accumulate_dx, void *stream) {
  VTC_CHECK(x && gamma && dy && dx && dgamma && dbeta && rows > 0, "layernorm_bwd: bad arguments");
  VTC_CHECK(width % 4 == 0 && width <= 1024, "layernorm_bwd: width=%d unsupported", width);
  (void)hipMemsetAsync(dgamma, 0, (size_t)width * 4, (hipStream_t)stream);
  (void)hipMemsetAsync(dbeta, 0, (size_t)width * 4, (hipStream_t)stream);
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, dgamma, dbeta, rows,
                     width, accumulate_dx);
  VTC_LAUNCH_CHECK("layernorm_bwd");
  return 0;
}

extern "C" int vtc_attention_small_bwd(const float *qkv, const float *dout, float *dqkv, int n_seq, int L, int heads, void *stream) {
  VTC_CHECK(qkv && dout && dqkv && n_seq > 0 && heads > 0, "attention_small_bwd: bad arguments");
  VTC_CHECK(L >= 1 && L <= 16, "attention_small_bwd: L=%d must be in [1, 16]", L);
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(cdiv(n_seq * heads, 4)), dim3(256), 0, (hipStream_t)stream, qkv, dout, dqkv, n_seq, L,
                     heads);
  VTC_LAUNCH_CHECK("attention_small_bwd");
  return 0;
}

extern "C" int vtc_quickgelu(const float *x, const float *dy, float *out, size_t n, void *stream) {
  VTC_CHECK(x && out && n > 0, "quickgelu: bad arguments");
  hipLaunchKernelGGL(quickgelu_kernel, GRID1(n), dim3(256), 0, (hipStream_t)stream, x, dy, out, n);
  VTC_LAUNCH_CHECK("quickgelu");
  return 0;
}

extern "C" int vtc_normalize_rows_bwd(const float *x, const float *dy, float *dx, int n, int d, void *stream) {
  VTC_CHECK(x && dy && dx && n > 0 && d > 0, "normalize_rows_bwd: bad arguments");
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3(cdiv(n, 4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, d);
  VTC_LAUNCH_CHECK("normalize_rows_bwd");
  return 0;
}

extern "C" int vtc_clip_loss_bwd(const float *sim, int n, float *dsim, void *ws, size_t ws_bytes, void *stream) {
  VTC_CHECK(sim && dsim && n > 0, "clip_loss_bwd: bad arguments");
  VTC_CHECK(ws && ws_bytes >= (size_t)4 * n * sizeof(float), "clip_loss_bwd: workspace too small");
  hipLaunchKernelGGL(lse_stats_kernel, dim3(cdiv(2 * n, 4)), dim3(256), 0, (hipStream_t)stream, sim, n, (float *)ws);
  hipLaunchKernelGGL(clip_loss_bwd_kernel, GRID1((size_t)n * n), dim3(256), 0, (hipStream_t)stream, sim, n, (const float *)ws, dsim);
  VTC_LAUNCH_CHECK("clip_loss_bwd");
  return 0;
}

extern "C" int vtc_adam_step(float *p, const float *g, float *m, float *v, float *vmax, size_t n, float lr, float beta1, float beta2,
                             float eps, int step, int amsgrad, void *stream) {
  VTC_CHECK(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  VTC_CHECK(!amsgrad || vmax, "adam_step: amsgrad needs vmax");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, GRID1(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, n, lr, beta1, beta2, eps, bc1,
                     sqrtf(bc2), amsgrad);
  VTC_LAUNCH_CHECK("adam_step");
  return 0;
}

extern "C" int vtc_axpby(float *out, const float *x, const float *y, float a, float b, size_t n, void *stream) {
  VTC_CHECK(out && x && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, GRID1(n), dim3(256), 0, (hipStream_t)stream, out, x, y, a, b, n);
  VTC_LAUNCH_CHECK("axpby");
  return 0;
}

extern "C" int vtc_scale_rows(float *x, const float *s, int rows, int d, int group, void *stream) {
  VTC_CHECK(x && s && rows > 0 && d > 0 && group > 0, "scale_rows: bad arguments");
  hipLaunchKernelGGL(scale_rows_kernel, GRID1((size_t)rows * d), dim3(256), 0, (hipStream_t)stream, x, s, rows, d, group);
  VTC_LAUNCH_CHECK("scale_rows");
  return 0;
}
