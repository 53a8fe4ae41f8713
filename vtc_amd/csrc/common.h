// Shared device/host helpers for libvtc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vtc_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VTC_WAVE 64

// Every kernel launch of the library is counted (one relaxed atomic add; vtc_debug_launch_count() reads the total): what
// bench.py reports as launches per forward at the small-batch operating points, where the launch count IS the cost.
void vtc_count_launch();
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)             \
  do {                                                                                              \
    vtc_count_launch();                                                                             \
    (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);            \
  } while (0)

void vtc_set_error(const char *fmt, ...);

#define VTC_CHECK(cond, ...)          \
  do {                                \
    if (!(cond)) {                    \
      vtc_set_error(__VA_ARGS__);     \
      return 1;                       \
    }                                 \
  } while (0)

#define VTC_LAUNCH_CHECK(name)                                                  \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      vtc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

// float -> bf16 (round to nearest even; NaN stays NaN via the hardware convert).
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
  return __builtin_bit_cast(float, ((unsigned)b) << 16);
}

// IEEE half as a second 16-bit operand format (VTC_F16): a distinct tag type so that templates can tell it from bf16.
// Same MFMA rate and bytes as bf16, 11 significant bits instead of 8, range +-65504 (the format upstream CLIP itself
// runs in on a GPU; clip.load(..., device="cuda") returns fp16 weights).
struct f16_t { unsigned short v; };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned short f2h(float f) {
  _Float16 h = (_Float16)f;   // v_cvt_f16_f32, round to nearest even
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float h2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// float -> raw 16 bits of the operand format T
template <typename T> __device__ __forceinline__ unsigned short cvt16(float f);
template <> __device__ __forceinline__ unsigned short cvt16<bf16_t>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ unsigned short cvt16<f16_t>(float f) { return f2h(f); }

// two floats -> one word of the operand format T, a in the low half (one v_cvt_pk_bf16_f32 / two v_cvt_f16_f32 + pack: same roundings as cvt16)
template <typename T> __device__ __forceinline__ unsigned pack16(float a, float b);
template <> __device__ __forceinline__ unsigned pack16<bf16_t>(float a, float b) {
  typedef float f2v_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf2v_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v_t){a, b}, bf2v_t));
}
template <> __device__ __forceinline__ unsigned pack16<f16_t>(float a, float b) {
  typedef float f2v_t __attribute__((ext_vector_type(2)));
  typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2v_t){a, b}, h2v_t));
}

// raw 16 bits of the operand format T -> float
template <typename T> __device__ __forceinline__ float up16(unsigned short b);
template <> __device__ __forceinline__ float up16<bf16_t>(unsigned short b) { return __builtin_bit_cast(float, ((unsigned)b) << 16); }
template <> __device__ __forceinline__ float up16<f16_t>(unsigned short b) { return h2f(b); }

template <typename T> struct ElemOps;
template <> struct ElemOps<float> {
  static constexpr int kDtype = VTC_F32;
  __device__ static __forceinline__ float load(const float *p) { return *p; }
  __device__ static __forceinline__ void store(float *p, float v) { *p = v; }
  __device__ static __forceinline__ void store4(float *p, float a, float b, float c, float d) {
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
  }
  __device__ static __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
};
template <> struct ElemOps<bf16_t> {
  static constexpr int kDtype = VTC_BF16;
  __device__ static __forceinline__ float load(const bf16_t *p) { return bf2f(*p); }
  __device__ static __forceinline__ void store(bf16_t *p, float v) { *p = f2bf(v); }
  __device__ static __forceinline__ void store4(bf16_t *p, float a, float b, float c, float d) {
    ushort4 v;
    v.x = f2bf(a); v.y = f2bf(b); v.z = f2bf(c); v.w = f2bf(d);
    *reinterpret_cast<ushort4 *>(p) = v;
  }
  __device__ static __forceinline__ float4 load4(const bf16_t *p) {
    ushort4 v = *reinterpret_cast<const ushort4 *>(p);
    return make_float4(bf2f(v.x), bf2f(v.y), bf2f(v.z), bf2f(v.w));
  }
};

template <> struct ElemOps<f16_t> {
  static constexpr int kDtype = VTC_F16;
  __device__ static __forceinline__ float load(const f16_t *p) { return h2f(p->v); }
  __device__ static __forceinline__ void store(f16_t *p, float v) { p->v = f2h(v); }
  __device__ static __forceinline__ void store4(f16_t *p, float a, float b, float c, float d) {
    ushort4 v;
    v.x = f2h(a); v.y = f2h(b); v.z = f2h(c); v.w = f2h(d);
    *reinterpret_cast<ushort4 *>(p) = v;
  }
  __device__ static __forceinline__ float4 load4(const f16_t *p) {
    ushort4 v = *reinterpret_cast<const ushort4 *>(p);
    return make_float4(h2f(v.x), h2f(v.y), h2f(v.z), h2f(v.w));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- per-device launch state ----------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting: a process that drives several GPUs (one
// thread per replica, train.py:77-80) must opt every device in.  One bit per device; setting it twice is harmless, so
// a race between two threads costs at most a redundant call.
#include <atomic>
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
};
// returns 0 on success; reports through vtc_set_error otherwise
inline int ensure_dynamic_lds(PerDeviceOnce &once, const void *kernel, int bytes, const char *name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const unsigned long long bit = 1ull << dev;
  if (once.done.load(std::memory_order_acquire) & bit) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    vtc_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", name, bytes, dev,
                  hipGetErrorString(e));
    return 1;
  }
  once.done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// ---- optional per-launch event timing (prof.hip) -------------------------------------------
struct ProfScope {
  ProfScope(int cls, double work, hipStream_t s, const int *m_dev = nullptr);
  ~ProfScope();
  void tag(int a, int b, int c);
  hipStream_t stream_;
  int idx_;
};

// marks every launch made while it is alive (on this thread) with a VTC_PROF_REGION_*
struct ProfRegion {
  explicit ProfRegion(int region);
  ~ProfRegion();
  int prev_;
};

// ---- internal launchers shared between translation units --------------------------------
struct GemmEpi {
  int mode = VTC_EPI_STORE;   // VTC_EPI_*  (+ internal modes below)
  int out_dtype = VTC_F32;
  int skip_mod = 0;
  // internal: patch-embed scatter (mode 3): m = (item*F + t)*P + n  ->  row item*T + 1 + n*F + t,
  //           out(fp32) = acc + pos[1+n] + temporal[t]
  const float *pos = nullptr;
  const float *temporal = nullptr;
  int P = 0, F = 0, T = 0;
  int frames_major = 0;      // 0: row item*T + 1 + n*F + t (timesformer_clip_alt.py:271-274); 1: item*T + 1 + t*P + n (timesformer_clip.py:392)
  // mode 3, gather = 1: A is the 16-bit PIXEL tensor [frames, 3, res, res] itself -- row m = patch (frame m / P, m % P), column
  // k = (c, i, j) of conv1's kernel (model/timesformer_clip_alt.py:262-263 `self.conv1(x)`, stride = kernel = patch) -- read in
  // place by the LDS-DMA source addressing: no im2row matrix.  patch in {16, 32}; the tensor is < 4 GiB.
  int gather = 0, grid = 0, res = 0, patch = 0;
  // internal: squared-L2 epilogue (mode 4): out = rown[m] + coln[n] - 2 acc
  const float *rown = nullptr;
  const float *coln = nullptr;
  // internal: scale by exp(*scale_log) (mode 5)
  const float *scale_log = nullptr;
  int ldo = 0;                // output leading dimension (0 => N)
  // internal: block-minima epilogue of the sweep (mode 6): no matrix is written.  Per (row, block of 64 columns): the three
  // smallest distance keys + the fourth smallest as a bound -> rowk[4][nblk_c][M]; with colk != nullptr also per (column,
  // block of RB rows, RB = the kernel's wave tile height) -> colk[4][nblk_r][N].
  // key = (bits(max(d, 0)) & ~127) | index in block: a non-negative float, compared as an unsigned integer.
  unsigned *rowk = nullptr, *colk = nullptr;
  int nblk_c = 0, nblk_r = 0, rb = 0;
  // internal: residual epilogue + the FOLLOWING LayerNorm (mode 7; 16-bit operands, N a multiple of 256, N <= 1024): the column
  // tile that finishes a 256-row block last normalises the block's rows (LN(out) * ln_g + ln_b -> ln_out, operand format).
  // ln_cnt: one arrival counter per row block, zeroed by the launcher.
  const float *ln_g = nullptr, *ln_b = nullptr;
  void *ln_out = nullptr;
  int *ln_cnt = nullptr;
  // LayerNorm folded into the GEMM that consumes it (16-bit operands; towers.hip "folded LayerNorm"):
  //   LN(x) W^T + b = rstd_m (x (g . W)^T - mean_m s_n) + c_n,   s_n = sum_k (g . W)_nk,   c_n = b_n + sum_k beta_k W_nk
  // producer (mode 2, the residual GEMM that writes x): y16 = x in the operand format [M, N] and fold_part[N / 64][M] =
  //   (sum, sum of squares) of the row's 64 columns of this wave -- every tile interior (M % 256 == 0, N % 256 == 0);
  // consumer (modes 0 / 1, 16-bit output): A = y16, W = (g . W) rounded, bias = c, fold_stat[M] = (mean, rstd), fold_s = s.
  void *y16 = nullptr;
  void *y16lo = nullptr;      // the residual stream is the pair (y16, y16lo): x = hi + lo, lo = fmt(x - hi) -- 16 + bits of
                              // mantissa in bf16, the same 8 bytes per element per residual GEMM as an fp32 stream, and hi IS
                              // the operand copy (no third array); `out` (fp32) is neither read nor written
  float *fold_part = nullptr;
  const float *fold_stat = nullptr, *fold_s = nullptr;
  // the row count M in DEVICE memory (*m_dev <= the M handed to launch_gemm, which then only sizes the grid): the ragged text
  // tower without a host sync (vtc_text_forward2) -- the persistent kernels read it when they start
  const int *m_dev = nullptr;
  // split-K (small M, long K: the MLP's c_proj at batch 1-2, whose 64 x 64 tiles are few and walk K = 3072 / 2048 alone): mode 0, fp32
  // out, 16-bit operands.  The K range is cut into `ksplit` slices; tile (m, n) of slice s writes its partial products to
  // out + s * split_stride (fp32 elements).  One tile per workgroup (the launcher checks that the chip holds them all): the sum over
  // the slices, the bias, the residual update and the LayerNorm statistics are norm.hip's splitk_resid_rows_kernel.
  int ksplit = 0, split_stride = 0;
};
enum { EPI_PATCH = 3, EPI_L2DIST = 4, EPI_SCALE = 5, EPI_L2MIN = 6, EPI_RESID_LN = 7,
       // modes 0 / 1 / 2 with the folded LayerNorm (fold_* fields), as instantiations of their own: chosen by launch_gemm
       EPI_FOLD_BASE = 8, EPI_STORE_FOLD = 8, EPI_GELU_FOLD = 9, EPI_RESID_FOLD = 10,
       EPI_RESID_FOLD_C = 11 /* ... that also re-centres the stream (fold_stat given) */,
       EPI_L2MIN2 = 12 /* EPI_L2MIN with TWO planes per block (the smallest key + the second as a bound): the recall-only sweep at small k */,
       EPI_L2MIN3 = 13 /* ... rows two planes, columns THREE (two keys + bound): small galleries, whose few, long column blocks make two planes' fp64 fallback frequent */ };
#define L2MIN_PLANES 4            // planes of EPI_L2MIN (three keys + bound); EPI_L2MIN2 / 3 write the first two / three plane slots of the same layout
constexpr int l2min_row_planes(int mode) { return (mode == EPI_L2MIN2 || mode == EPI_L2MIN3) ? 2 : L2MIN_PLANES; }     // per (row, block of 64 columns)
constexpr int l2min_col_planes(int mode) { return mode == EPI_L2MIN2 ? 2 : (mode == EPI_L2MIN3 ? 3 : L2MIN_PLANES); }  // per (column, block of RB rows)
constexpr bool l2min_half_keys(int mode) { return mode == EPI_L2MIN2 || mode == EPI_L2MIN3; }     // keys = half distances from pre-loaded accumulators (gemm.hip)

int launch_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                const GemmEpi &epi, hipStream_t stream);
bool gemm_resid_ln_supported(int M, int N, int K, int dtype);
bool gemm_patch_gather_supported(int n_frames, int grid, int patch, int res, int pixel_dtype, int dtype);
// rows_dev != NULL: the row count lives in device memory (*rows_dev <= rows, which then only sizes the grid)
int launch_layernorm(const float *x, const float *g, const float *b, void *y, int rows, int width, int out_dtype,
                     const int *row_index, int row_mul, bool no_norm, hipStream_t stream, const int *rows_dev = nullptr);
int launch_fold_stats(const float *part, int nb, int rows, float *stat, hipStream_t stream, const int *rows_dev = nullptr);
int launch_ln_cast_rowstats(const float *x, const float *gamma, const float *beta, void *y16, void *y16lo, float *stat, int rows, int width, int dtype,
                            hipStream_t stream);
int launch_splitk_resid_rows(const float *part, int nsl, int stride, const float *bias, void *hi, void *lo, float *stat, int rows, int width, int dtype,
                             hipStream_t stream, const int *rows_dev = nullptr);
int launch_cast_rowstats(const float *x, void *y16, void *y16lo, float *stat, int rows, int width, int dtype, hipStream_t stream,
                         const int *rows_dev = nullptr);
// dst[i] = src row (row_index[i], or i * row_mul): rows of row_bytes (a multiple of 16) bytes
int launch_gather_rows(const void *src, void *dst, int n, int row_bytes, const int *row_index, int row_mul, hipStream_t stream);
int launch_split_merge_rows(const void *hi, const void *lo, float *x, int n, int width, const int *row_index, int row_mul, int dtype,
                            hipStream_t stream, const int *rows_dev = nullptr);
// token ids of a text-tower call: sequences [0, n_a) are rows of `a`, [n_a, n_a + n_b) rows of `b` (titles + comments without a
// concatenated copy; b may be NULL with n_b = 0)
struct TextIds {
  const int64_t *a, *b;
  int n_a, n_b;
  __host__ __device__ const int64_t *row(int s, int ctx) const { return s < n_a ? a + (size_t)s * ctx : b + (size_t)(s - n_a) * ctx; }
};
int launch_single_query_attention(const void *qkv, const void *q, float *out, int n_out, int L, int heads, int s2, int a0, int a1, int a2,
                                  int a3, int pstride, const int *eot, const int *offs, int ctx, int dtype, hipStream_t stream);
int launch_mean_cast(const float *x, void *out, int n, int F, int W, int dtype, hipStream_t stream);
int launch_attention(const void *qkv, void *out, float *cls_out, int n_seq, int L, int heads, int causal, int s2,
                     int a0, int a1, int a2, int a3, int pstride, int dtype, hipStream_t stream);
