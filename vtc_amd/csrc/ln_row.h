// ln_row.h -- the arithmetic of one LayerNorm row (model/timesformer_clip_alt.py:22-28: fp32 compute, nn.LayerNorm's eps
// 1e-5 and biased variance), shared by the stand-alone kernel (norm.hip) and the residual GEMM's fused LayerNorm
// (gemm.hip, EPI_RESID_LN) so that both produce the same bits.
// One wave per row; lane l owns the 8-column chunks l, l + 64 (LN_MAXV chunks: width <= 1024), preloaded into v[i][0..1].
#pragma once
#include "common.h"

constexpr int LN_MAXV = 2;

template <typename OutT, bool NO_NORM>
__device__ __forceinline__ void ln_row_compute(const float4 (&v)[LN_MAXV][2], const float *__restrict__ gamma,
                                               const float *__restrict__ beta, OutT *yr, int width, int lane) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width)
      s += ((v[i][0].x + v[i][0].y) + (v[i][0].z + v[i][0].w)) + ((v[i][1].x + v[i][1].y) + (v[i][1].z + v[i][1].w));
  }
  float mean = 0.f, rstd = 1.f;
  if (!NO_NORM) {
    mean = wave_sum(s) / width;
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
      }
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / width + 1e-5f);  // nn.LayerNorm default eps, biased variance
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < width) {
      float o[8] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
      if (!NO_NORM) {
        const float4 g0 = *reinterpret_cast<const float4 *>(gamma + c), g1 = *reinterpret_cast<const float4 *>(gamma + c + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(beta + c), b1 = *reinterpret_cast<const float4 *>(beta + c + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (o[e] - mean) * rstd * gm[e] + bt[e];
      }
      if constexpr (sizeof(OutT) == 2) {
        uint4 pk;
        pk.x = (unsigned)cvt16<OutT>(o[0]) | ((unsigned)cvt16<OutT>(o[1]) << 16);
        pk.y = (unsigned)cvt16<OutT>(o[2]) | ((unsigned)cvt16<OutT>(o[3]) << 16);
        pk.z = (unsigned)cvt16<OutT>(o[4]) | ((unsigned)cvt16<OutT>(o[5]) << 16);
        pk.w = (unsigned)cvt16<OutT>(o[6]) | ((unsigned)cvt16<OutT>(o[7]) << 16);
        *reinterpret_cast<uint4 *>(yr + c) = pk;
      } else {
        ElemOps<OutT>::store4(yr + c, o[0], o[1], o[2], o[3]);
        ElemOps<OutT>::store4(yr + c + 4, o[4], o[5], o[6], o[7]);
      }
    }
  }
}
