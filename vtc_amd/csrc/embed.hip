// embed.hip -- data-movement kernels around the towers (all HBM-bound, 16-byte accesses):
//   im2row          pixels [N,3,H,W] -> patch rows [N*P, 3*p*p]   (conv1 as a GEMM,
//                   model/timesformer_clip_alt.py:255-260 / upstream VisionTransformer.conv1)
//   cls_rows        x[item*T] = class_embedding + pos[0]           (:262-269)
//   cls_mean        space-attention cls output = mean over frames  (:162-164)
//   text_embed      token_embedding[ids] + positional_embedding, and the EOT row index
//                   ids.argmax(-1)                                  (upstream CLIP.encode_text)
//   cam_tokens / cam_finalize   Context Adapter Module glue         (model/model.py:150-151,157-159,203,208-212)
#include "common.h"

namespace {

// raw uint8 pixels: ToTensor (/255) + per-channel Normalize fused into the gather
struct PixNorm { float mean[3], inv_std[3]; };
template <typename T>
__global__ __launch_bounds__(256) void im2row_u8_kernel(const unsigned char *__restrict__ px, T *__restrict__ out, int n_frames, int grid,
                                                        int patch, int res, PixNorm nrm) {
  const int K = 3 * patch * patch;
  const size_t total = (size_t)n_frames * grid * grid * (K / 4);
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kq = (int)(idx % (K / 4));
  const size_t m = idx / (K / 4);
  const int k = kq * 4;
  const int c = k / (patch * patch), rem = k - c * patch * patch;
  const int i = rem / patch, j = rem - i * patch;
  const int P = grid * grid;
  const int f = (int)(m / P), pp = (int)(m - (size_t)f * P);
  const int py = pp / grid, pxx = pp - py * grid;
  const unsigned char *src = px + (((size_t)f * 3 + c) * res + (size_t)py * patch + i) * res + (size_t)pxx * patch + j;
  const uchar4 u = *reinterpret_cast<const uchar4 *>(src);
  const float mu = nrm.mean[c], is = nrm.inv_std[c];
  ElemOps<T>::store4(out + m * K + k, (u.x / 255.0f - mu) * is, (u.y / 255.0f - mu) * is, (u.z / 255.0f - mu) * is,
                     (u.w / 255.0f - mu) * is);
}

// raw uint8 pixels -> the same [frames, 3, res, res] tensor in the 16-bit operand format, ToTensor (/255) + per-channel Normalize
// of CLIP_TRANSFORM (dataset_loaders/dataset_loaders.py:40-49) applied on the way: what the patch-gather GEMM (gemm.hip, EPI_PATCH
// with `gather`) reads in place through its LDS-DMA source addressing.  16 pixels per thread: one 16-byte load, two 16-byte
// stores (a plane is a multiple of 16 pixels, so a thread never straddles two channels).
template <typename T>
__global__ __launch_bounds__(256) void pixels_u8_to_operand_kernel(const unsigned char *__restrict__ px, T *__restrict__ out, size_t n16,
                                                                   int plane16, PixNorm nrm) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n16) return;
  const int c = (int)((idx / plane16) % 3);
  const float mu = nrm.mean[c], is = nrm.inv_std[c];
  const uint4 u = *reinterpret_cast<const uint4 *>(px + idx * 16);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
  unsigned o[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = ((w[q] & 255u) / 255.0f - mu) * is, b = (((w[q] >> 8) & 255u) / 255.0f - mu) * is;
    const float cc = (((w[q] >> 16) & 255u) / 255.0f - mu) * is, d = ((w[q] >> 24) / 255.0f - mu) * is;
    o[2 * q] = (unsigned)cvt16<T>(a) | ((unsigned)cvt16<T>(b) << 16);
    o[2 * q + 1] = (unsigned)cvt16<T>(cc) | ((unsigned)cvt16<T>(d) << 16);
  }
  uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned short *>(out) + idx * 16);
  dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
  dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

template <typename PixT, typename T>
__global__ __launch_bounds__(256) void im2row_kernel(const PixT *__restrict__ px, T *__restrict__ out, int n_frames, int grid,
                                                     int patch, int res) {
  // one thread = 4 consecutive k (same image row segment); k = c*p*p + i*p + j
  const int K = 3 * patch * patch;
  const size_t total = (size_t)n_frames * grid * grid * (K / 4);
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kq = (int)(idx % (K / 4));
  const size_t m = idx / (K / 4);
  const int k = kq * 4;
  const int c = k / (patch * patch), rem = k - c * patch * patch;
  const int i = rem / patch, j = rem - i * patch;
  const int P = grid * grid;
  const int f = (int)(m / P), pp = (int)(m - (size_t)f * P);
  const int py = pp / grid, pxx = pp - py * grid;
  const PixT *src = px + (((size_t)f * 3 + c) * res + (size_t)py * patch + i) * res + (size_t)pxx * patch + j;
  const float4 v = ElemOps<PixT>::load4(src);
  ElemOps<T>::store4(out + m * K + k, v.x, v.y, v.z, v.w);
}

// Any patch size (ViT-L/14: patch 14, K = 588): one thread per output element, K padded with zeros to `kpad` columns (the GEMM
// consumes K in 128-byte rows; the packed conv1 weight is zero-padded to the same width).  PixT = unsigned char applies
// ToTensor + Normalize on the way.
template <typename PixT, typename T>
__global__ __launch_bounds__(256) void im2row_generic_kernel(const PixT *__restrict__ px, T *__restrict__ out, int n_frames, int grid,
                                                             int patch, int res, int kpad, PixNorm nrm) {
  const int K = 3 * patch * patch;
  const size_t total = (size_t)n_frames * grid * grid * kpad;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % kpad);
  const size_t m = idx / kpad;
  float v = 0.f;
  if (k < K) {
    const int c = k / (patch * patch), rem = k - c * patch * patch;
    const int i = rem / patch, j = rem - i * patch;
    const int P = grid * grid;
    const int f = (int)(m / P), pp = (int)(m - (size_t)f * P);
    const int py = pp / grid, pxx = pp - py * grid;
    const size_t src = (((size_t)f * 3 + c) * res + (size_t)py * patch + i) * res + (size_t)pxx * patch + j;
    if constexpr (sizeof(PixT) == 1) v = ((float)px[src] / 255.0f - nrm.mean[c]) * nrm.inv_std[c];
    else v = ElemOps<PixT>::load(px + src);
  }
  ElemOps<T>::store(out + idx, v);
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float *x, const float *__restrict__ cls, const float *__restrict__ pos0,
                                                       int n_items, int T, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_items * W) return;
  const int item = i / W, c = i - item * W;
  x[(size_t)item * T * W + c] = cls[c] + pos0[c];
}

template <typename T>
__global__ __launch_bounds__(256) void cls_mean_kernel(const float *__restrict__ cls_tmp, T *__restrict__ out, int n_items,
                                                       int F, int Ttok, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_items * W) return;
  const int item = i / W, c = i - item * W;
  float s = 0.f;
  for (int t = 0; t < F; ++t) s += cls_tmp[((size_t)item * F + t) * W + c];
  ElemOps<T>::store(out + (size_t)item * Ttok * W + c, s / F);
}

__global__ __launch_bounds__(256) void text_embed_kernel(const TextIds ids, const float *__restrict__ tok,
                                                         const float *__restrict__ pos, float *__restrict__ x, int n_rows,
                                                         int ctx, int W, int vocab) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int sq = r / ctx, p = r - sq * ctx;
  long id = ids.row(sq, ctx)[p];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // never fault on a bad id
  const float *tr = tok + (size_t)id * W, *pr = pos + (size_t)p * W;
  float *xr = x + (size_t)r * W;
  for (int c = lane * 4; c < W; c += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(tr + c), b = *reinterpret_cast<const float4 *>(pr + c);
    *reinterpret_cast<float4 *>(xr + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// Ragged text batch: only the tokens up to and including EOT are materialised (a causal tower never
// lets a later position influence the EOT feature).  Row seq_offsets[s] + p <- token p of sequence s.
__global__ __launch_bounds__(256) void text_embed_ragged_kernel(const TextIds ids, const float *__restrict__ tok,
                                                                const float *__restrict__ pos, const int *__restrict__ seq_offsets,
                                                                float *__restrict__ x, int n_seq, int ctx, int W, int vocab) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);          // dense (s, p) index
  if (r >= n_seq * ctx) return;
  const int s = r / ctx, p = r - s * ctx;
  const int lo = seq_offsets[s], len = seq_offsets[s + 1] - lo;
  if (p >= len) return;
  long id = ids.row(s, ctx)[p];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const float *tr = tok + (size_t)id * W, *pr = pos + (size_t)p * W;
  float *xr = x + (size_t)(lo + p) * W;
  for (int c = lane * 4; c < W; c += 256) {
    const float4 a = *reinterpret_cast<const float4 *>(tr + c), b = *reinterpret_cast<const float4 *>(pr + c);
    *reinterpret_cast<float4 *>(xr + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ __launch_bounds__(256) void last_row_kernel(const int *__restrict__ seq_offsets, int *__restrict__ rows, int n_seq) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n_seq) rows[s] = seq_offsets[s + 1] - 1;            // the EOT row of sequence s
}

// eot_row[s] = s*ctx + argmax_p ids[s,p] (first maximum, as torch.argmax)
__global__ __launch_bounds__(256) void eot_index_kernel(const TextIds ids, int *__restrict__ eot_row, int n_seq, int ctx) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= n_seq) return;
  const int64_t *r = ids.row(s, ctx);
  int64_t best = r[0];
  int bi = 0;
  for (int p = 1; p < ctx; ++p)
    if (r[p] > best) { best = r[p]; bi = p; }
  eot_row[s] = s * ctx + bi;
}

// ---- ragged text batch without host knowledge of the lengths (vtc_text_forward2): what `text.argmax(-1)` (upstream
// CLIP.encode_text: the EOT id 49407 is the largest), a cumsum and a D2H of the total did on the host in round 2 -----------
// lens[s] = (position of the first maximum id) + 1.  One wave per sequence.
__global__ __launch_bounds__(256) void seq_len_kernel(const TextIds ids, int *__restrict__ lens, int n_seq, int ctx) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_seq) return;
  const int64_t *r = ids.row(s, ctx);
  long long best = (long long)0x8000000000000000ull;
  int bi = 0x7fffffff;
  for (int p = lane; p < ctx; p += 64) {
    const long long v = r[p];
    if (v > best) { best = v; bi = p; }          // ascending p per lane: the first maximum of the lane
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) lens[s] = bi + 1;
}
// offsets[0..n_seq] = exclusive prefix sums of lens; m_dev[0] = total rows, m_dev[1] = total rounded up to 256.  One workgroup.
__global__ __launch_bounds__(1024) void scan_offsets_kernel(const int *__restrict__ lens, int *__restrict__ offsets, int n_seq,
                                                            int *__restrict__ m_dev) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_seq; base += 1024) {
    const int i = base + tid;
    const int v = i < n_seq ? lens[i] : 0;
    int inc = v;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const int carry = carry_s;
    if (i < n_seq) offsets[i] = carry + wbase + inc - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wbase + inc;
    __syncthreads();
  }
  if (tid == 0) {
    const int total = carry_s;
    offsets[n_seq] = total;
    m_dev[0] = total;
    m_dev[1] = (total + 255) / 256 * 256;
  }
}

// X[b*Lc + 0] = normalize(main[b]);  X[b*Lc + 1 + c] = normalize(empty(b,c) ? mask : comm[b*nc + c])
__global__ __launch_bounds__(256) void cam_tokens_kernel(const float *__restrict__ main_f, const float *__restrict__ comm,
                                                         const int64_t *__restrict__ comments, const float *__restrict__ mask_emb,
                                                         float *__restrict__ X, int B, int nc, int ctx, int D) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Lc = 1 + nc;
  if (r >= B * Lc) return;
  const int b = r / Lc, t = r - b * Lc;
  const float *src;
  if (t == 0) src = main_f + (size_t)b * D;
  else {
    const int ci = b * nc + (t - 1);
    const bool empty = comments[(size_t)ci * ctx + 1] == 49407;   // model/model.py:208
    src = empty ? mask_emb : comm + (size_t)ci * D;
  }
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += src[c] * src[c];
  const float nrm = sqrtf(wave_sum(s));
  for (int c = lane; c < D; c += 64) X[(size_t)r * D + c] = src[c] / nrm;
}

__device__ __forceinline__ float act_apply(int act, float v, float msq, float scale) {
  // model/model.py:34-39,65-77; msq = sum((s+1e-9)^2) for squash, sum((s+1e-9)^2) for normalize
  if (act == VTC_ACT_NORMALIZE) return (v + 1e-9f) / sqrtf(msq);
  if (act == VTC_ACT_SQUASH) {
    const float mag = sqrtf(msq);
    return scale * (msq / (1.0f + msq)) * ((v + 1e-9f) / mag);
  }
  if (act == VTC_ACT_TANH) return tanhf(v);
  return v;
}

// one wave per item: r = init_from_avg ? normalize(mean_i normalize(Y_i)) : lin[b];
// r = act(r); adapted = normalize(normalize(main) + r)
template <int MAXD64>
__global__ __launch_bounds__(256) void cam_finalize_kernel(const float *__restrict__ Y, const float *__restrict__ lin,
                                                           const float *__restrict__ main_f, float *__restrict__ out, int B, int Lc,
                                                           int D, int init_from_avg, int act, float scale,
                                                           const float *__restrict__ bn_mean, const float *__restrict__ bn_var) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float r[MAXD64];
#pragma unroll
  for (int k = 0; k < MAXD64; ++k) r[k] = 0.f;
  if (init_from_avg) {
    for (int t = 0; t < Lc; ++t) {
      const float *y = Y + ((size_t)b * Lc + t) * D;
      float v[MAXD64], s = 0.f;
#pragma unroll
      for (int k = 0; k < MAXD64; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < D ? y[c] : 0.f;
        s += v[k] * v[k];
      }
      const float nrm = sqrtf(wave_sum(s));
#pragma unroll
      for (int k = 0; k < MAXD64; ++k) r[k] += v[k] / nrm;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) { r[k] /= Lc; s += r[k] * r[k]; }
    const float nrm = sqrtf(wave_sum(s));
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) r[k] /= nrm;
  } else {
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) {
      const int c = lane + 64 * k;
      r[k] = c < D ? lin[(size_t)b * D + c] : 0.f;
    }
  }
  if (act == VTC_ACT_SUB_MEAN || act == VTC_ACT_BN) {
    // eval-mode BatchNorm1d(affine=False): model/model.py:42-61 (running statistics, eps 1e-5)
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) {
      const int c = lane + 64 * k;
      if (c < D) {
        const float d = r[k] - bn_mean[c];
        r[k] = act == VTC_ACT_BN ? d / sqrtf(bn_var[c] + 1e-5f) : d;
      }
    }
  } else if (act != VTC_ACT_NONE) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) {
      const int c = lane + 64 * k;
      if (c < D) s += (r[k] + 1e-9f) * (r[k] + 1e-9f);
    }
    const float msq = wave_sum(s);
#pragma unroll
    for (int k = 0; k < MAXD64; ++k) r[k] = act_apply(act, r[k], msq, scale);
  }
  float m[MAXD64], s = 0.f;
#pragma unroll
  for (int k = 0; k < MAXD64; ++k) {
    const int c = lane + 64 * k;
    m[k] = c < D ? main_f[(size_t)b * D + c] : 0.f;
    s += m[k] * m[k];
  }
  const float mn = sqrtf(wave_sum(s));
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < MAXD64; ++k) {
    const int c = lane + 64 * k;
    m[k] = c < D ? m[k] / mn + r[k] : 0.f;
    s2 += m[k] * m[k];
  }
  const float an = sqrtf(wave_sum(s2));
#pragma unroll
  for (int k = 0; k < MAXD64; ++k) {
    const int c = lane + 64 * k;
    if (c < D) out[(size_t)b * D + c] = m[k] / an;
  }
}

}  // namespace

// columns of the patch matrix / of the packed conv1 weight: 3 patch^2 rounded up to whole 128-byte GEMM rows of either operand size
int patch_k_padded(int patch) { return (3 * patch * patch + 63) / 64 * 64; }

int launch_im2row(const void *px, int pixel_dtype, void *out, int dtype, int n_frames, int grid, int patch, int res,
                  const float *mean, const float *stdv, hipStream_t stream) {
  if (patch % 4 != 0 || patch_k_padded(patch) != 3 * patch * patch) {
    // patch 14 (ViT-L/14, model/timesformer_clip_alt.py:304-310): element-wise gather, zero-padded K
    const int kpad = patch_k_padded(patch);
    PixNorm nrm = {{0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    if (pixel_dtype == VTC_U8)
      for (int c = 0; c < 3; ++c) {
        VTC_CHECK(stdv[c] > 0.f, "im2row: pix_std[%d] must be positive for uint8 pixels", c);
        nrm.mean[c] = mean[c]; nrm.inv_std[c] = 1.0f / stdv[c];
      }
    const size_t total = (size_t)n_frames * grid * grid * kpad;
    const dim3 g((unsigned)((total + 255) / 256)), b(256);
    ProfScope prof(VTC_PROF_EMBED, (double)total * ((pixel_dtype == VTC_U8 ? 1 : pixel_dtype == VTC_BF16 ? 2 : 4) + (dtype == VTC_F32 ? 4 : 2)), stream);
    VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "im2row: output dtype %d", dtype);
#define VTC_IM2ROW_G(PT, OT) hipLaunchKernelGGL((im2row_generic_kernel<PT, OT>), g, b, 0, stream, (const PT *)px, (OT *)out, n_frames, grid, patch, res, kpad, nrm)
#define VTC_IM2ROW_GO(PT) do { if (dtype == VTC_BF16) VTC_IM2ROW_G(PT, bf16_t); else if (dtype == VTC_F16) VTC_IM2ROW_G(PT, f16_t); else VTC_IM2ROW_G(PT, float); } while (0)
    if (pixel_dtype == VTC_U8) VTC_IM2ROW_GO(unsigned char);
    else if (pixel_dtype == VTC_F32) VTC_IM2ROW_GO(float);
    else if (pixel_dtype == VTC_F16) VTC_IM2ROW_GO(f16_t);
    else VTC_IM2ROW_GO(bf16_t);
#undef VTC_IM2ROW_GO
#undef VTC_IM2ROW_G
    VTC_LAUNCH_CHECK("im2row_generic");
    return 0;
  }
  if (pixel_dtype == VTC_U8) {
    PixNorm nrm;
    for (int c = 0; c < 3; ++c) {
      VTC_CHECK(stdv[c] > 0.f, "im2row: pix_std[%d] must be positive for uint8 pixels", c);
      nrm.mean[c] = mean[c]; nrm.inv_std[c] = 1.0f / stdv[c];
    }
    const size_t total8 = (size_t)n_frames * grid * grid * (3 * patch * patch / 4);
    ProfScope prof8(VTC_PROF_EMBED, (double)total8 * 4 * (1 + (dtype == VTC_BF16 ? 2 : 4)), stream);
    if (dtype == VTC_BF16)
      hipLaunchKernelGGL((im2row_u8_kernel<bf16_t>), dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, stream,
                         (const unsigned char *)px, (bf16_t *)out, n_frames, grid, patch, res, nrm);
    else if (dtype == VTC_F16)
      hipLaunchKernelGGL((im2row_u8_kernel<f16_t>), dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, stream,
                         (const unsigned char *)px, (f16_t *)out, n_frames, grid, patch, res, nrm);
    else
      hipLaunchKernelGGL((im2row_u8_kernel<float>), dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, stream,
                         (const unsigned char *)px, (float *)out, n_frames, grid, patch, res, nrm);
    VTC_LAUNCH_CHECK("im2row_u8");
    return 0;
  }
  const size_t total = (size_t)n_frames * grid * grid * (3 * patch * patch / 4);
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  ProfScope prof(VTC_PROF_EMBED, (double)total * 4 * ((pixel_dtype == VTC_BF16 ? 2 : 4) + (dtype == VTC_BF16 ? 2 : 4)), stream);
#define VTC_IM2ROW(PT, OT) hipLaunchKernelGGL((im2row_kernel<PT, OT>), g, b, 0, stream, (const PT *)px, (OT *)out, n_frames, grid, patch, res)
  if (dtype == VTC_F16) {        // IEEE-half operand mode of the vision towers (round 6)
    if (pixel_dtype == VTC_F32) VTC_IM2ROW(float, f16_t);
    else if (pixel_dtype == VTC_F16) VTC_IM2ROW(f16_t, f16_t);
    else VTC_IM2ROW(bf16_t, f16_t);
  } else if (pixel_dtype == VTC_F16) {
    if (dtype == VTC_BF16) VTC_IM2ROW(f16_t, bf16_t);
    else VTC_IM2ROW(f16_t, float);
  } else
#undef VTC_IM2ROW
  if (pixel_dtype == VTC_F32 && dtype == VTC_BF16)
    hipLaunchKernelGGL((im2row_kernel<float, bf16_t>), g, b, 0, stream, (const float *)px, (bf16_t *)out, n_frames, grid, patch, res);
  else if (pixel_dtype == VTC_F32)
    hipLaunchKernelGGL((im2row_kernel<float, float>), g, b, 0, stream, (const float *)px, (float *)out, n_frames, grid, patch, res);
  else if (dtype == VTC_BF16)
    hipLaunchKernelGGL((im2row_kernel<bf16_t, bf16_t>), g, b, 0, stream, (const bf16_t *)px, (bf16_t *)out, n_frames, grid, patch, res);
  else
    hipLaunchKernelGGL((im2row_kernel<bf16_t, float>), g, b, 0, stream, (const bf16_t *)px, (float *)out, n_frames, grid, patch, res);
  VTC_LAUNCH_CHECK("im2row");
  return 0;
}

// uint8 pixels [n_frames, 3, res, res] -> normalised pixels in the 16-bit operand format `dtype`, same layout
int launch_pixels_u8_to_operand(const void *px, void *out, int dtype, int n_frames, int res, const float *mean, const float *stdv,
                                hipStream_t stream) {
  VTC_CHECK((dtype == VTC_BF16 || dtype == VTC_F16) && (res * res) % 16 == 0 && ((uintptr_t)px & 15) == 0,
            "pixels_u8_to_operand: dtype %d, resolution %d, pixel pointer alignment", dtype, res);
  PixNorm nrm;
  for (int c = 0; c < 3; ++c) { nrm.mean[c] = mean[c]; nrm.inv_std[c] = 1.0f / stdv[c]; }
  const size_t n16 = (size_t)n_frames * 3 * res * res / 16;
  ProfScope prof(VTC_PROF_EMBED, (double)n16 * 48, stream);
  if (dtype == VTC_BF16)
    hipLaunchKernelGGL((pixels_u8_to_operand_kernel<bf16_t>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream,
                       (const unsigned char *)px, (bf16_t *)out, n16, res * res / 16, nrm);
  else
    hipLaunchKernelGGL((pixels_u8_to_operand_kernel<f16_t>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream,
                       (const unsigned char *)px, (f16_t *)out, n16, res * res / 16, nrm);
  VTC_LAUNCH_CHECK("pixels_u8_to_operand");
  return 0;
}

int launch_cls_rows(float *x, const float *cls, const float *pos0, int n_items, int T, int W, hipStream_t stream) {
  hipLaunchKernelGGL(cls_rows_kernel, dim3(cdiv(n_items * W, 256)), dim3(256), 0, stream, x, cls, pos0, n_items, T, W);
  VTC_LAUNCH_CHECK("cls_rows");
  return 0;
}

int launch_cls_mean(const float *cls_tmp, void *out, int dtype, int n_items, int F, int T, int W, hipStream_t stream) {
  VTC_CHECK(dtype == VTC_BF16 || dtype == VTC_F32 || dtype == VTC_F16, "cls_mean: dtype %d", dtype);
  const dim3 g(cdiv(n_items * W, 256)), b(256);
  if (dtype == VTC_BF16) hipLaunchKernelGGL((cls_mean_kernel<bf16_t>), g, b, 0, stream, cls_tmp, (bf16_t *)out, n_items, F, T, W);
  else if (dtype == VTC_F16) hipLaunchKernelGGL((cls_mean_kernel<f16_t>), g, b, 0, stream, cls_tmp, (f16_t *)out, n_items, F, T, W);
  else hipLaunchKernelGGL((cls_mean_kernel<float>), g, b, 0, stream, cls_tmp, (float *)out, n_items, F, T, W);
  VTC_LAUNCH_CHECK("cls_mean");
  return 0;
}

// ---- token packing (dataset_loaders/dataset_loaders.py:224-248 `_tokenise`, its array-building half: the BPE
// encoder and the RAKE summariser in front of it are host text processing) ------------------------------------------------------
// ids[s] = [SOT] + tokens[offsets[s] : offsets[s + 1]] + [EOT], zero padded to ctx; a sequence whose SOT + tokens + EOT reach ctx
// keeps its first ctx - 1 ids and ends in EOT (:240-243 `tokens[: max_len - 1] + [eot_token]`).  One thread per output id.
__global__ __launch_bounds__(256) void pack_tokens_kernel(const int *__restrict__ tokens, const int *__restrict__ offsets, int n_seq, int ctx,
                                                          int sot, int eot, int64_t *__restrict__ ids) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n_seq * ctx) return;
  const int s = (int)(idx / ctx), p = (int)(idx - (size_t)s * ctx);
  const int lo = offsets[s], n = offsets[s + 1] - lo;
  const int len = min(n + 2, ctx);                       // packed length, EOT included
  int64_t v = 0;
  if (p == 0) v = sot;
  else if (p == len - 1) v = eot;
  else if (p < len - 1) v = tokens[lo + p - 1];
  ids[idx] = v;
}

int launch_text_prep(const TextIds &ids, int n_seq, int ctx, int *lens, int *offsets, int *m_dev, hipStream_t stream) {
  ProfScope prof(VTC_PROF_EMBED, (double)n_seq * ctx * 8, stream);
  hipLaunchKernelGGL(seq_len_kernel, dim3(cdiv(n_seq, 4)), dim3(256), 0, stream, ids, lens, n_seq, ctx);
  hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(1024), 0, stream, lens, offsets, n_seq, m_dev);
  VTC_LAUNCH_CHECK("text_prep");
  return 0;
}

int launch_text_embed(const TextIds &ids, const float *tok, const float *pos, float *x, int *eot_row, int n_seq, int ctx,
                      int W, int vocab, hipStream_t stream) {
  VTC_CHECK(W % 4 == 0, "text_embed: width %d", W);
  ProfScope prof(VTC_PROF_EMBED, (double)n_seq * ctx * W * 12, stream);
  hipLaunchKernelGGL(text_embed_kernel, dim3(cdiv(n_seq * ctx, 4)), dim3(256), 0, stream, ids, tok, pos, x, n_seq * ctx, ctx, W, vocab);
  hipLaunchKernelGGL(eot_index_kernel, dim3(cdiv(n_seq, 256)), dim3(256), 0, stream, ids, eot_row, n_seq, ctx);
  VTC_LAUNCH_CHECK("text_embed");
  return 0;
}

int launch_text_embed_ragged(const TextIds &ids, const float *tok, const float *pos, const int *seq_offsets, float *x, int *eot_row,
                             int n_seq, int ctx, int W, int vocab, hipStream_t stream) {
  VTC_CHECK(W % 4 == 0, "text_embed: width %d", W);
  ProfScope prof(VTC_PROF_EMBED, (double)n_seq * ctx * W * 6, stream);
  hipLaunchKernelGGL(text_embed_ragged_kernel, dim3(cdiv(n_seq * ctx, 4)), dim3(256), 0, stream, ids, tok, pos, seq_offsets, x, n_seq,
                     ctx, W, vocab);
  hipLaunchKernelGGL(last_row_kernel, dim3(cdiv(n_seq, 256)), dim3(256), 0, stream, seq_offsets, eot_row, n_seq);
  VTC_LAUNCH_CHECK("text_embed_ragged");
  return 0;
}

int launch_cam_tokens(const float *main_f, const float *comm, const int64_t *comments, const float *mask_emb, float *X, int B,
                      int nc, int ctx, int D, hipStream_t stream) {
  hipLaunchKernelGGL(cam_tokens_kernel, dim3(cdiv(B * (1 + nc), 4)), dim3(256), 0, stream, main_f, comm, comments, mask_emb, X, B,
                     nc, ctx, D);
  VTC_LAUNCH_CHECK("cam_tokens");
  return 0;
}

int launch_cam_finalize(const float *Y, const float *lin, const float *main_f, float *out, int B, int Lc, int D, int init_from_avg,
                        int act, float scale, const float *bn_mean, const float *bn_var, hipStream_t stream) {
  VTC_CHECK(D <= 1024, "cam: width %d > 1024", D);
  VTC_CHECK(act >= VTC_ACT_NONE && act <= VTC_ACT_BN, "cam: unknown residual activation %d", act);
  VTC_CHECK((act != VTC_ACT_SUB_MEAN && act != VTC_ACT_BN) || (bn_mean && bn_var),
            "cam: residual activation %d needs bn_mean / bn_var", act);
  hipLaunchKernelGGL((cam_finalize_kernel<16>), dim3(cdiv(B, 4)), dim3(256), 0, stream, Y, lin, main_f, out, B, Lc, D, init_from_avg,
                     act, scale, bn_mean, bn_var);
  VTC_LAUNCH_CHECK("cam_finalize");
  return 0;
}

extern "C" int vtc_pack_tokens(const int *tokens, const int *offsets, int n_seq, int ctx, int sot, int eot, int64_t *ids, void *stream) {
  VTC_CHECK(offsets && ids && n_seq > 0 && ctx >= 2, "pack_tokens: bad arguments (n_seq=%d ctx=%d)", n_seq, ctx);
  hipLaunchKernelGGL(pack_tokens_kernel, dim3(cdiv(n_seq * ctx, 256)), dim3(256), 0, (hipStream_t)stream, tokens, offsets, n_seq, ctx, sot, eot, ids);
  VTC_LAUNCH_CHECK("pack_tokens");
  return 0;
}
