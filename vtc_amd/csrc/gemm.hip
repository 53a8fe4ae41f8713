// gemm.hip -- the one GEMM of the hot path:  out[M,N] = epi(A[M,K] @ W[N,K]^T + bias)
//
// Used for every projection of the towers (QKV, out-proj, temporal_fc, c_fc, c_proj, patch
// embedding, output projections, the CAM blocks), for the batch similarity and for the N x N
// distance matrix of the retrieval sweep.  Replaces the cuBLAS/MKL GEMMs behind
// model/timesformer_clip_alt.py:50,65,148,174 and upstream nn.MultiheadAttention / nn.Linear.
//
// gfx950 design
//   * two tile configurations of one kernel template:
//       "big"   256x256 output tile, 512 threads (8 waves as 2x4, 128x64 per wave), one workgroup
//               per CU, 128 KiB LDS -- 131 FLOP per byte staged from L2, used when the problem
//               has enough tiles to fill the chip;
//       "small" 128x128 tile, 256 threads (4 waves as 2x2, 64x64 per wave), two workgroups per CU
//               -- for the skinny problems (CAM, output projections, edge cases) and for fp32;
//   * K is consumed in 128-byte rows (64 bf16 / 32 fp32 per step); both operands go L2 -> LDS with
//     global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip), double buffered, issued from inline
//     asm so the compiler does not serialise the pipeline behind them; one s_waitcnt vmcnt(0) +
//     barrier per K-step publishes the next buffer;
//   * LDS image is lane-linear (a DMA constraint); the XOR bank swizzle is applied on the per-lane
//     SOURCE address and again on the ds_read_b128 address (guide rule 21): conflict-free reads;
//   * bf16: v_mfma_f32_16x16x32_bf16; fp32: 4 x v_mfma_f32_16x16x4_f32 per 16-byte chunk (exact
//     fp32; the k order inside a chunk is permuted identically for both operands);
//   * the MFMA "A" operand is the WEIGHT fragment and "B" the ACTIVATION fragment, so a lane ends
//     up holding 4 consecutive output columns of one output row: bias / QuickGELU / residual /
//     scatter epilogues run straight from the accumulators with 16-byte (fp32) or 8-byte (bf16)
//     accesses, no LDS round trip, all loads of a row issued before their first use;
//   * PERSISTENT workgroups: the grid is one (big) or two (small) workgroups per CU; each walks a
//     strided list of tiles, and the first K-slab of the NEXT tile is already in flight during the
//     last K-step of the current one, so short-K problems (K = 512/768 here) do not pay a cold
//     prologue per tile;
//   * workgroup -> tile map is XCD-aware: workgroups with equal (id mod 8) share an XCD (observed
//     dispatch; used for speed only) and walk a contiguous range of tiles ordered in super-rows of
//     1024 output rows x all columns, so the weight panels and the activation panels in flight stay
//     resident in that XCD's 4 MiB L2.
#include "common.h"

namespace {

constexpr int ROWB = 128;          // bytes of K per LDS row
constexpr int SUPER_ROWS = 1024;   // output rows per L2 super-row

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

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB).
// Inline asm so that hipcc does NOT model it as a memory operation: with the builtin the compiler
// waits vmcnt(0) before the first ds_read of every K-step (it cannot prove the DMA into the other
// buffer does not alias the reads) and the pipeline serialises.  The completion wait is ours: one
// s_waitcnt vmcnt(0) before the barrier that publishes the buffer.  M0 carries the wave-uniform LDS
// byte address and is saved/restored inside the same statement (guide 5.7).
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

// Stage GROUPS 8-row groups of a ROWS-row x 128-byte operand tile (this wave's share).
template <int GROUPS>
__device__ __forceinline__ void stage_tile(const char *base, int row0, int nrows, int ld_bytes, int kbyte, unsigned lds_tile,
                                           int wave, int lane) {
#pragma unroll
  for (int q = 0; q < GROUPS; ++q) {
    const int group = wave * GROUPS + q;      // 8-row group
    const int r = group * 8 + (lane >> 3);    // tile row this lane fills
    const int cs = lane & 7;                  // LDS chunk slot (linear)
    const int c = cs ^ ((r >> 1) & 7);        // source chunk (swizzle on the source side)
    int gr = row0 + r;
    gr = gr < nrows ? gr : nrows - 1;         // clamp: tail rows re-read a valid row, never stored
    const char *src = base + (size_t)gr * ld_bytes + kbyte + c * 16;
    glds16(src, lds_tile + group * 1024);
  }
}

// QuickGELU x * sigmoid(1.702 x) (model/timesformer_clip_alt.py:31-33).  fp32 mode: IEEE division and
// expf; bf16 mode: v_exp_f32 + v_rcp_f32 (1 ulp each, far below the bf16 rounding of the result) --
// the IEEE division sequence alone cost ~25 % of a K = 512 tile.
template <bool ACCURATE>
__device__ __forceinline__ float quick_gelu(float x) {
  if (ACCURATE) return x / (1.0f + expf(-1.702f * x));
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554670f * x));   // 1.702 * log2(e)
}

// WM x WN waves, each owning TM x TN MFMA tiles of 16x16.
template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM *WN * 64, 2) void gemm_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
  constexpr int AG = BM / 8 / NW, WG = BN / 8 / NW;
  constexpr int SUPER = SUPER_ROWS / BM;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int g = lane >> 4;

  // ---- persistent, XCD-aware tile walk ----------------------------------------------------
  const int ntiles = p.MT * p.NT, nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int nb_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);                 // workgroups sharing this XCD label
  const int nt_x = (ntiles >> 3) + (xcd < (ntiles & 7) ? 1 : 0);           // tiles given to this XCD label
  const int start_x = xcd * (ntiles >> 3) + min(xcd, ntiles & 7);
  auto decode = [&](int logical, int &m0, int &n0) {
    const int per_super = SUPER * p.NT;
    const int sr = logical / per_super, rem = logical - sr * per_super;
    const int gsz = min(SUPER, p.MT - sr * SUPER);
    const int nt = rem / gsz;
    m0 = (sr * SUPER + (rem - nt * gsz)) * BM;
    n0 = nt * BN;
  };
  int li = slot;
  if (li >= nt_x) return;                      // uniform for the whole workgroup
  int m0, n0;
  decode(start_x + li, m0, n0);

  const int ksteps = p.K / Mma<T>::KPR;
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void *)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int swz = (lane >> 1) & 7;
  const int arow = (wr * TM * 16 + (lane & 15)) * ROWB;
  const int wrow = (wc * TN * 16 + (lane & 15)) * ROWB;
  const int ldo = p.ldo;
  const bool vec_ok = (ldo & 3) == 0;

  stage_tile<AG>(p.A, m0, p.M, p.lda_bytes, 0, lds_base, wave_u, lane);
  stage_tile<WG>(p.W, n0, p.N, p.ldw_bytes, 0, lds_base + A_BYTES, wave_u, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;

  while (true) {
    const int li_next = li + nb_x;
    const bool has_next = li_next < nt_x;
    int m0n = 0, n0n = 0;
    if (has_next) decode(start_x + li_next, m0n, n0n);

    // Per-column addend of the epilogue (bias, or |g|^2 for the distance epilogue), fetched now so
    // its latency hides under the K loop.  Interior tiles only; edge tiles use the generic path.
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && vec_ok;
    float4 pre4[TN];
    {
      const float *colv = MODE == EPI_L2DIST ? p.epi.coln : p.bias;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        pre4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (interior && colv) pre4[j] = *reinterpret_cast<const float4 *>(colv + n0 + (wc * TN + j) * 16 + g * 4);
      }
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < ksteps; ++t) {
      const unsigned nxt = lds_base + (cur ^ 1) * STAGE;
      if (t + 1 < ksteps) {
        stage_tile<AG>(p.A, m0, p.M, p.lda_bytes, (t + 1) * ROWB, nxt, wave_u, lane);
        stage_tile<WG>(p.W, n0, p.N, p.ldw_bytes, (t + 1) * ROWB, nxt + A_BYTES, wave_u, lane);
      } else if (has_next) {                   // first K-slab of the next tile rides under this step
        stage_tile<AG>(p.A, m0n, p.M, p.lda_bytes, 0, nxt, wave_u, lane);
        stage_tile<WG>(p.W, n0n, p.N, p.ldw_bytes, 0, nxt + A_BYTES, wave_u, lane);
      }
      const char *as = lds + cur * STAGE;
      const char *ws = as + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int coff = ((4 * ks + g) ^ swz) << 4;
        uint4 af[TM], wf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4 *>(ws + wrow + j * 16 * ROWB + coff);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4 *>(as + arow + i * 16 * ROWB + coff);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], af[i], acc[i][j]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = m0 + 16 (wr TM + i) + (lane & 15),
    //                n = n0 + 16 (wc TN + j) + 4 g
    float scale = 1.0f;
    if (MODE == EPI_SCALE) scale = __expf(*p.epi.scale_log);
    if (interior) {
      // Fast path (every tile of the towers): straight-line code, all loads of a row issued
      // before the first use so a row costs one memory round trip.
      const int nb = n0 + wc * TN * 16 + g * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wr * TM + i) * 16 + (lane & 15);
        if (MODE == VTC_EPI_RESID && p.epi.skip_mod > 0 && (m % p.epi.skip_mod) == 0) continue;
        size_t orow = (size_t)m;
        float4 add4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) add4[j] = pre4[j];
        if (MODE == EPI_PATCH) {
          const int np = m % p.epi.P, ft = m / p.epi.P;
          const int tt = ft % p.epi.F, item = ft / p.epi.F;
          orow = (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
          const float *posrow = p.epi.pos + (size_t)(1 + np) * p.N + nb;
#pragma unroll
          for (int j = 0; j < TN; ++j) add4[j] = *reinterpret_cast<const float4 *>(posrow + 16 * j);
          if (p.epi.temporal) {
            const float *temprow = p.epi.temporal + (size_t)tt * p.N + nb;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float4 t4 = *reinterpret_cast<const float4 *>(temprow + 16 * j);
              add4[j].x += t4.x; add4[j].y += t4.y; add4[j].z += t4.z; add4[j].w += t4.w;
            }
          }
        }
        float rn = 0.f;
        if (MODE == EPI_L2DIST) rn = p.epi.rown[m];
        OutT *o = reinterpret_cast<OutT *>(p.out) + orow * ldo + nb;
        float4 x4[TN];
        if (MODE == VTC_EPI_RESID) {
#pragma unroll
          for (int j = 0; j < TN; ++j) x4[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<float *>(o) + 16 * j);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          const float a4[4] = {add4[j].x, add4[j].y, add4[j].z, add4[j].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (MODE == EPI_L2DIST) v[e] = rn + a4[e] - 2.0f * v[e];
            else v[e] += a4[e];
            if (MODE == VTC_EPI_GELU) v[e] = quick_gelu<sizeof(T) == 4>(v[e]);
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
    } else {
      // Generic path: edge tiles (M or N not a multiple of the tile, odd leading dimension).
      // Fully unrolled: a runtime index into acc[][] would send the accumulators to scratch.
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wr * TM + i) * 16 + (lane & 15);
        bool live = m < p.M;
        if (MODE == VTC_EPI_RESID && p.epi.skip_mod > 0 && live && (m % p.epi.skip_mod) == 0) live = false;
        if (live) {
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
          for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wc * TN + j) * 16 + g * 4;
            OutT *o = reinterpret_cast<OutT *>(p.out) + orow * ldo + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e < p.N) {
                float x = acc[i][j][e];
                if (p.bias) x += p.bias[n + e];
                if (MODE == VTC_EPI_GELU) x = quick_gelu<sizeof(T) == 4>(x);
                if (MODE == EPI_PATCH) x += posrow[n + e] + (temprow ? temprow[n + e] : 0.f);
                if (MODE == EPI_L2DIST) x = rn + p.epi.coln[n + e] - 2.0f * x;
                if (MODE == EPI_SCALE) x *= scale;
                if (MODE == VTC_EPI_RESID) reinterpret_cast<float *>(o)[e] += x;
                else ElemOps<OutT>::store(o + e, x);
              }
            }
          }
        }
      }
    }

    if (!has_next) break;
    li = li_next; m0 = m0n; n0 = n0n;
  }
}

int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  return g_num_cus;
}

template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN>
int run(GemmParams p, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT_ = WM * WN * 64;
  p.MT = cdiv(p.M, BM); p.NT = cdiv(p.N, BN);
  const int ntiles = p.MT * p.NT;
  const size_t shmem = (size_t)2 * (BM + BN) * ROWB;
  const int wg_per_cu = shmem > 80 * 1024 ? 1 : 2;
  const int grid = min(ntiles, num_cus() * wg_per_cu);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<T, MODE, OutT, WM, WN, TM, TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, MODE, OutT, WM, WN, TM, TN>), dim3(grid), dim3(NT_), shmem, stream, p);
  VTC_LAUNCH_CHECK("gemm");
  return 0;
}

int g_force_tile = 0;   // 0 = heuristic, 1 = small, 2 = big (diagnostics: VTC_GEMM_TILE)

template <typename T, int MODE, typename OutT>
int run_cfg(const GemmParams &p, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    // big tiles when they still fill the chip ~1.5 times over
    const long big_tiles = (long)cdiv(p.M, 256) * cdiv(p.N, 256);
    bool big = big_tiles * 2 >= (long)num_cus() * 3;
    if (g_force_tile == 1) big = false;
    if (g_force_tile == 2) big = true;
    if (big) return run<T, MODE, OutT, 2, 4, 8, 4>(p, stream);
  }
  return run<T, MODE, OutT, 2, 2, 4, 4>(p, stream);
}

template <typename T>
int dispatch(const GemmParams &p, hipStream_t stream) {
  const bool out_f32 = p.epi.out_dtype == VTC_F32;
  switch (p.epi.mode) {
    case VTC_EPI_STORE:
      if (out_f32) return run_cfg<T, VTC_EPI_STORE, float>(p, stream);
      return run_cfg<T, VTC_EPI_STORE, bf16_t>(p, stream);
    case VTC_EPI_GELU:
      if (out_f32) return run_cfg<T, VTC_EPI_GELU, float>(p, stream);
      return run_cfg<T, VTC_EPI_GELU, bf16_t>(p, stream);
    case VTC_EPI_RESID: return run_cfg<T, VTC_EPI_RESID, float>(p, stream);
    case EPI_PATCH: return run_cfg<T, EPI_PATCH, float>(p, stream);
    case EPI_L2DIST: return run_cfg<T, EPI_L2DIST, float>(p, stream);
    case EPI_SCALE: return run_cfg<T, EPI_SCALE, float>(p, stream);
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
  static bool env_done = false;
  if (!env_done) {
    const char *e = getenv("VTC_GEMM_TILE");
    if (e) g_force_tile = atoi(e);
    env_done = true;
  }
  GemmParams p;
  p.A = (const char *)A; p.W = (const char *)W; p.bias = bias; p.out = out;
  p.M = M; p.N = N; p.K = K;
  p.lda_bytes = K * esz; p.ldw_bytes = K * esz;
  p.ldo = epi.ldo > 0 ? epi.ldo : N;
  p.MT = 0; p.NT = 0;
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
