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

// ---- optional per-launch event timing (prof.hip) -------------------------------------------
struct ProfScope {
  ProfScope(int cls, double work, hipStream_t s);
  ~ProfScope();
  hipStream_t stream_;
  int idx_;
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
  // internal: squared-L2 epilogue (mode 4): out = rown[m] + coln[n] - 2 acc
  const float *rown = nullptr;
  const float *coln = nullptr;
  // internal: scale by exp(*scale_log) (mode 5)
  const float *scale_log = nullptr;
  int ldo = 0;                // output leading dimension (0 => N)
};
enum { EPI_PATCH = 3, EPI_L2DIST = 4, EPI_SCALE = 5 };

int launch_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                const GemmEpi &epi, hipStream_t stream);
int launch_layernorm(const float *x, const float *g, const float *b, void *y, int rows, int width, int out_dtype,
                     const int *row_index, int row_mul, bool no_norm, hipStream_t stream);
int launch_attention(const void *qkv, void *out, float *cls_out, int n_seq, int L, int heads, int causal, int s2,
                     int a0, int a1, int a2, int a3, int pstride, int dtype, hipStream_t stream);
