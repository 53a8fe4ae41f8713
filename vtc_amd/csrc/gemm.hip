// gemm.hip -- the one GEMM of the hot path:  out[M,N] = epi(A[M,K] @ W[N,K]^T + bias)
//
// Used for every projection of the towers (QKV, out-proj, temporal_fc, c_fc, c_proj, patch
// embedding, output projections, the CAM blocks), for the batch similarity and for the N x N
// distance matrix of the retrieval sweep.  Replaces the cuBLAS/MKL GEMMs behind
// model/timesformer_clip_alt.py:50,65,148,174 and upstream nn.MultiheadAttention / nn.Linear.
//
// gfx950 design
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
//     tiles of 16x16), K consumed in 128-byte rows (64 bf16 / 32 fp32 per step);
//   * both operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double
//     buffered; LDS image is lane-linear, the XOR bank swizzle is applied on the per-lane SOURCE
//     address and again on the ds_read_b128 address (guide rule 21);
//   * bf16: v_mfma_f32_16x16x32_bf16; fp32: 4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk (exact
//     fp32, k order inside a chunk permuted identically for both operands);
//   * the MFMA "A" operand is the WEIGHT fragment and "B" the ACTIVATION fragment, so a lane ends
//     up holding 4 consecutive output columns of one output row: bias / QuickGELU / residual /
//     scatter epilogues run straight from the accumulators with 16-byte (fp32) or 8-byte (bf16)
//     accesses, no LDS round trip;
//   * workgroup ids are remapped so that each XCD walks a contiguous range of tiles in
//     (8 row-tiles x all column-tiles) super-rows: the weight panel and 8 activation panels in
//     flight stay resident in that XCD's 4 MiB L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                 // bytes of K per LDS row
constexpr int TILE_BYTES = BM * ROWB;     // 16 KiB per operand per stage
constexpr int NTHREADS = 256;
constexpr int SUPER = 8;                  // row tiles per L2 super-row

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int KPR = 64;  // K elements per 128-byte row
  __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc,
                                                  0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int KPR = 32;
  __device__ static __forceinline__ void run(const uint4 &w, const uint4 &a, f32x4 &acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.x), __builtin_bit_cast(float, a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.y), __builtin_bit_cast(float, a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.z), __builtin_bit_cast(float, a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, w.w), __builtin_bit_cast(float, a.w), acc, 0, 0, 0);
  }
};

struct GemmParams {
  const char *A;
  const char *W;
  const float *bias;
  void *out;
  int M, N, K;
  int lda_bytes, ldw_bytes, ldo;
  int MT, NT;
  GemmEpi epi;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB).
// Written as inline asm so that hipcc does NOT model it as a memory operation: with the builtin the
// compiler serialises the pipeline by waiting vmcnt(0) before the first ds_read of every K-step
// (it cannot prove the DMA into the other buffer does not alias the reads).  The completion wait is
// ours: one s_waitcnt vmcnt(0) before the barrier that publishes the buffer.  M0 carries the
// wave-uniform LDS byte address and is saved/restored inside the same statement (guide 5.7).
__device__ __forceinline__ void glds16(const char *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Stage one 128-row x 128-byte operand tile: 16 wave-instructions of 1 KiB, 4 per wave.
__device__ __forceinline__ void stage_tile(const char *base, int row0, int nrows, int ld_bytes, int kbyte, unsigned lds_tile,
                                           int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int group = wave * 4 + q;           // 8-row group
    const int r = group * 8 + (lane >> 3);    // tile row this lane fills
    const int cs = lane & 7;                  // LDS chunk slot (linear)
    const int c = cs ^ ((r >> 1) & 7);        // source chunk (swizzle on the source side)
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;         // clamp: tail rows re-read a valid row, never stored
    const char *src = base + (size_t)gr * ld_bytes + kbyte + c * 16;
    glds16(src, lds_tile + group * 1024);
  }
}

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

template <typename T, int MODE, typename OutT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // ---- XCD-aware tile walk -------------------------------------------------------------
  int mt, nt;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_super = SUPER * p.NT;
    const int sr = logical / per_super, rem = logical - sr * per_super;
    const int g = min(SUPER, p.MT - sr * SUPER);
    nt = rem / g;
    mt = sr * SUPER + (rem - nt * g);
  }
  const int m0 = mt * BM, n0 = nt * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Per-column addend of the epilogue (bias, or |g|^2 for the distance epilogue), fetched now so
  // that its latency hides under the K loop.  Interior tiles only; edge tiles use the slow path.
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && ((p.ldo & 3) == 0);
  float4 pre4[4];
  {
    const float *colv = MODE == EPI_L2DIST ? p.epi.coln : p.bias;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pre4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (interior && colv) pre4[j] = *reinterpret_cast<const float4 *>(colv + n0 + wc * 64 + j * 16 + (lane >> 4) * 4);
    }
  }

  const int ksteps = p.K / Mma<T>::KPR;
  // wave-uniform LDS byte address of the staging area (dynamic LDS starts at the kernel's LDS base)
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // prologue
  stage_tile(p.A, m0, p.M, p.lda_bytes, 0, lds_base, wave_u, lane);
  stage_tile(p.W, n0, p.N, p.ldw_bytes, 0, lds_base + TILE_BYTES, wave_u, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA landed (this wave's share)
  __syncthreads();

  const int g = lane >> 4;
  const int swz = (lane >> 1) & 7;
  const int arow = (wr * 64 + (lane & 15)) * ROWB;
  const int wrow = (wc * 64 + (lane & 15)) * ROWB;

  int cur = 0;
  for (int t = 0; t < ksteps; ++t) {
    if (t + 1 < ksteps) {
      const unsigned nxt = lds_base + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile(p.A, m0, p.M, p.lda_bytes, (t + 1) * ROWB, nxt, wave_u, lane);
      stage_tile(p.W, n0, p.N, p.ldw_bytes, (t + 1) * ROWB, nxt + TILE_BYTES, wave_u, lane);
    }
    const char *as = lds + cur * 2 * TILE_BYTES;
    const char *ws = as + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((4 * ks + g) ^ swz) << 4;
      uint4 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4 *>(as + arow + i * 16 * ROWB + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const uint4 *>(ws + wrow + j * 16 * ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<T>::run(wf[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds out[m][n..n+3], m = m0 + 64 wr + 16 i + (lane & 15),
  //                n = n0 + 64 wc + 16 j + 4 g
  const int ldo = p.ldo;
  const bool vec_ok = (ldo & 3) == 0;
  float scale = 1.0f;
  if (MODE == EPI_SCALE) scale = __expf(*p.epi.scale_log);

  if (interior) {
    // Fast path (every tile of the towers): straight-line code, all loads of a row issued before
    // the first use so the epilogue costs one memory round trip instead of sixteen.
    const int nb = n0 + wc * 64 + g * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wr * 64 + i * 16 + (lane & 15);
      if (MODE == VTC_EPI_RESID && p.epi.skip_mod > 0 && (m % p.epi.skip_mod) == 0) continue;
      size_t orow = (size_t)m;
      float4 add4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) add4[j] = pre4[j];
      if (MODE == EPI_PATCH) {
        const int np = m % p.epi.P, ft = m / p.epi.P;
        const int tt = ft % p.epi.F, item = ft / p.epi.F;
        orow = (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
        const float *posrow = p.epi.pos + (size_t)(1 + np) * p.N + nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) add4[j] = *reinterpret_cast<const float4 *>(posrow + 16 * j);
        if (p.epi.temporal) {
          const float *temprow = p.epi.temporal + (size_t)tt * p.N + nb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 t4 = *reinterpret_cast<const float4 *>(temprow + 16 * j);
            add4[j].x += t4.x; add4[j].y += t4.y; add4[j].z += t4.z; add4[j].w += t4.w;
          }
        }
      }
      float rn = 0.f;
      if (MODE == EPI_L2DIST) rn = p.epi.rown[m];
      OutT *o = reinterpret_cast<OutT *>(p.out) + orow * ldo + nb;
      float4 x4[4];
      if (MODE == VTC_EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x4[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<float *>(o) + 16 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const float a4[4] = {add4[j].x, add4[j].y, add4[j].z, add4[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (MODE == EPI_L2DIST) v[e] = rn + a4[e] - 2.0f * v[e];
          else v[e] += a4[e];
          if (MODE == VTC_EPI_GELU) v[e] = quick_gelu(v[e]);
          if (MODE == EPI_SCALE) v[e] *= scale;
        }
        if (MODE == VTC_EPI_RESID) {
          *reinterpret_cast<float4 *>(reinterpret_cast<float *>(o) + 16 * j) =
              make_float4(x4[j].x + v[0], x4[j].y + v[1], x4[j].z + v[2], x4[j].w + v[3]);
        } else {
          ElemOps<OutT>::store4(o + 16 * j, v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }

  // Generic path: edge tiles (M or N not a multiple of 128, odd leading dimension).
  // (fully unrolled: a runtime index into acc[][] would send the accumulators to scratch)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + (lane & 15);
    if (m >= p.M) continue;
    if (MODE == VTC_EPI_RESID && p.epi.skip_mod > 0 && (m % p.epi.skip_mod) == 0) continue;
    size_t orow = (size_t)m;
    const float *posrow = nullptr, *temprow = nullptr;
    float rn = 0.f;
    if (MODE == EPI_PATCH) {
      const int np = m % p.epi.P, ft = m / p.epi.P;
      const int tt = ft % p.epi.F, item = ft / p.epi.F;
      orow = (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
      posrow = p.epi.pos + (size_t)(1 + np) * p.N;
      if (p.epi.temporal) temprow = p.epi.temporal + (size_t)tt * p.N;
    }
    if (MODE == EPI_L2DIST) rn = p.epi.rown[m];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + g * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      OutT *o = reinterpret_cast<OutT *>(p.out) + orow * ldo + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= p.N) continue;
        float x = v[e];
        if (p.bias) x += p.bias[n + e];
        if (MODE == VTC_EPI_GELU) x = quick_gelu(x);
        if (MODE == EPI_PATCH) x += posrow[n + e] + (temprow ? temprow[n + e] : 0.f);
        if (MODE == EPI_L2DIST) x = rn + p.epi.coln[n + e] - 2.0f * x;
        if (MODE == EPI_SCALE) x *= scale;
        if (MODE == VTC_EPI_RESID) reinterpret_cast<float *>(o)[e] += x;
        else ElemOps<OutT>::store(o + e, x);
      }
    }
  }
}

template <typename T, int MODE, typename OutT>
int run(const GemmParams &p, hipStream_t stream) {
  const int grid = p.MT * p.NT;
  const size_t shmem = 4 * TILE_BYTES;  // 64 KiB: 2 stages x (A + W)
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<T, MODE, OutT>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, MODE, OutT>), dim3(grid), dim3(NTHREADS), shmem, stream, p);
  VTC_LAUNCH_CHECK("gemm");
  return 0;
}

template <typename T>
int dispatch(const GemmParams &p, hipStream_t stream) {
  const bool out_f32 = p.epi.out_dtype == VTC_F32;
  switch (p.epi.mode) {
    case VTC_EPI_STORE:
      if (out_f32) return run<T, VTC_EPI_STORE, float>(p, stream);
      return run<T, VTC_EPI_STORE, bf16_t>(p, stream);
    case VTC_EPI_GELU:
      if (out_f32) return run<T, VTC_EPI_GELU, float>(p, stream);
      return run<T, VTC_EPI_GELU, bf16_t>(p, stream);
    case VTC_EPI_RESID: return run<T, VTC_EPI_RESID, float>(p, stream);
    case EPI_PATCH: return run<T, EPI_PATCH, float>(p, stream);
    case EPI_L2DIST: return run<T, EPI_L2DIST, float>(p, stream);
    case EPI_SCALE: return run<T, EPI_SCALE, float>(p, stream);
  }
  vtc_set_error("gemm: unknown epilogue %d", p.epi.mode);
  return 1;
}

}  // namespace

int launch_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                const GemmEpi &epi, hipStream_t stream) {
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16, "gemm: bad dtype %d", dtype);
  const int esz = dtype == VTC_BF16 ? 2 : 4;
  const int kpr = ROWB / esz;
  VTC_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  VTC_CHECK(K % kpr == 0, "gemm: K=%d must be a multiple of %d for dtype %d", K, kpr, dtype);
  VTC_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, "gemm: operands must be 16-byte aligned");
  VTC_CHECK(!(epi.out_dtype == VTC_BF16 && (epi.mode == VTC_EPI_RESID || epi.mode >= EPI_PATCH)),
            "gemm: epilogue %d writes fp32 only", epi.mode);
  GemmParams p;
  p.A = (const char *)A; p.W = (const char *)W; p.bias = bias; p.out = out;
  p.M = M; p.N = N; p.K = K;
  p.lda_bytes = K * esz; p.ldw_bytes = K * esz;
  p.ldo = epi.ldo > 0 ? epi.ldo : N;
  p.MT = cdiv(M, BM); p.NT = cdiv(N, BN);
  p.epi = epi;
  ProfScope prof(dtype == VTC_BF16 ? VTC_PROF_GEMM_BF16 : VTC_PROF_GEMM_F32, 2.0 * M * N * K, stream);
  return dtype == VTC_BF16 ? dispatch<bf16_t>(p, stream) : dispatch<float>(p, stream);
}

extern "C" int vtc_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
                        int epilogue, int out_dtype, int skip_mod, void *stream) {
  VTC_CHECK(epilogue >= VTC_EPI_STORE && epilogue <= VTC_EPI_RESID, "vtc_gemm: bad epilogue %d", epilogue);
  GemmEpi e;
  e.mode = epilogue; e.out_dtype = epilogue == VTC_EPI_RESID ? VTC_F32 : out_dtype; e.skip_mod = skip_mod;
  return launch_gemm(A, W, bias, out, M, N, K, dtype, e, (hipStream_t)stream);
}
