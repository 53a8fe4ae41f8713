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
#include "gemm_common.h"

using namespace vtcgemm;

namespace {

// WM x WN waves, each owning TM x TN MFMA tiles of 16x16.
// NSTAGE LDS buffers: 2 = wait for the next slab at the end of every K-step; 3 = the LDS-DMA of slab
// t+2 is issued at step t and only slab t+1 is waited for (counted s_waitcnt vmcnt(G)), so two slabs are
// always in flight and the DMA latency has two K-steps of matrix work to hide under.
template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(WM *WN * 64, 2) void gemm_kernel(const GemmParams p) {
  constexpr int NW = WM * WN, BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int G = BM / 8 / NW + BN / 8 / NW;          // LDS-DMA instructions per wave per slab
  constexpr bool STAGGER = NW == 8;                     // two waves per SIMD inside one workgroup
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

  if (p.exp_arg > 0 && bid >= (nwg >> 1)) {   // diagnostic: phase-shift the second workgroup of each CU
    for (int i = 0; i < p.exp_arg; ++i) __builtin_amdgcn_s_sleep(16);
  }
  int cur = 0;                                   // LDS buffer of the slab being multiplied
  // both operands addressable with 32-bit byte offsets (true for every tower shape)?
  const bool small32 = (size_t)p.M * p.lda_bytes < (1ull << 32) && (size_t)p.N * p.ldw_bytes < (1ull << 32);
  unsigned a_off[AG], w_off[WG];
  tile_offsets<AG>(a_off, m0, p.M, p.lda_bytes, wave_u, lane);
  tile_offsets<WG>(w_off, n0, p.N, p.ldw_bytes, wave_u, lane);
  int m0n = 0, n0n = 0;
  bool has_next = li + nb_x < nt_x;
  if (has_next) decode(start_x + li + nb_x, m0n, n0n);
  // stage the slab that is `d` K-steps after step t of the current tile (it may belong to the next tile)
  auto stage_ahead = [&](int t, int d) -> bool {
    const unsigned dst = lds_base + ((cur + d) % NSTAGE) * STAGE;
    const int k = t + d;
    if (k < ksteps) {
      if (small32) {                             // uniform base + precomputed 32-bit lane offsets
        stage_tile_fast<AG>(a_off, p.A + (size_t)k * ROWB, dst, wave_u);
        stage_tile_fast<WG>(w_off, p.W + (size_t)k * ROWB, dst + A_BYTES, wave_u);
      } else {
        stage_tile<AG>(p.A, m0, p.M, p.lda_bytes, k * ROWB, dst, wave_u, lane);
        stage_tile<WG>(p.W, n0, p.N, p.ldw_bytes, k * ROWB, dst + A_BYTES, wave_u, lane);
      }
      return true;
    }
    if (has_next) {                              // rides under the last K-steps of this tile
      stage_tile<AG>(p.A, m0n, p.M, p.lda_bytes, (k - ksteps) * ROWB, dst, wave_u, lane);
      stage_tile<WG>(p.W, n0n, p.N, p.ldw_bytes, (k - ksteps) * ROWB, dst + A_BYTES, wave_u, lane);
      return true;
    }
    return false;
  };
  stage_ahead(0, 0);
  if (NSTAGE == 3 && stage_ahead(0, 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  while (true) {

    // Interior tiles (every tile of the towers) take the transposed fast epilogue; edge tiles the generic one.
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N) && vec_ok;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < ksteps; ++t) {
      // Stagger (MI355X_MICROARCH "two waves per SIMD", item 9): the two waves that share a SIMD run the
      // same program between the same barriers, so left alone they issue their LDS-DMA pieces (~100
      // cycles of issue each, no matrix work) at the same time and their MFMAs at the same time.  The
      // second half of the workgroup issues its DMA after the first half-step instead, so one wave's
      // address/DMA issue sits beside the other's MFMAs.
      const bool late_dma = STAGGER && wave_u >= NW / 2;
      bool issued = false;
      if (!late_dma) issued = stage_ahead(t, NSTAGE - 1);
      // Fragment reads are inline-asm ds_read_b128 with hand-counted lgkmcnt waits: hipcc, left to itself,
      // keeps ONE A-fragment register in this loop (read -> lgkmcnt(0) -> 4 MFMA -> read ...), which parks
      // every wave on an LDS round trip per 4 MFMAs (SQ_WAIT_ANY 42 %, MFMA busy 44 %).  Here the read of
      // fragment i+1 is in flight while the MFMAs of fragment i issue (LDS returns in order, so
      // lgkmcnt(1) == "everything but the youngest read has landed").
      const unsigned a_addr = lds_base + cur * STAGE + arow;
      const unsigned w_addr = lds_base + cur * STAGE + A_BYTES + wrow;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (ks == 1 && late_dma) issued = stage_ahead(t, NSTAGE - 1);
        const unsigned coff = ((4 * ks + g) ^ swz) << 4;
        u32x4 wf[TN], aE, aO;
#pragma unroll
        for (int j = 0; j < TN; ++j) lds_read16(wf[j], w_addr + coff, j * 16 * ROWB);
        lds_read16(aE, a_addr + coff, 0);
#pragma unroll
        for (int i = 0; i < TM; i += 2) {
          lds_read16(aO, a_addr + coff, (i + 1) * 16 * ROWB);
          if (i == 0) lgkm_wait_frags<1>(aE, wf);
          else lgkm_wait<1>(aE);
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aE, acc[i][j]);
          if (i + 2 < TM) {
            lds_read16(aE, a_addr + coff, (i + 2) * 16 * ROWB);
            lgkm_wait<1>(aO);
          } else {
            lgkm_wait<0>(aO);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], aO, acc[i + 1][j]);
        }
      }
      // the next slab must have landed; with 3 stages the slab issued in this step may stay in flight
      if (NSTAGE == 3 && issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }

    // ---- epilogue: lane holds out[m][n..n+3], m = m0 + 16 (wr TM + i) + (lane & 15),
    //                n = n0 + 16 (wc TN + j) + 4 g
    float scale = 1.0f;
    if (MODE == EPI_SCALE) scale = __expf(*p.epi.scale_log);
    if (interior) {
      // Fast path (every tile of the towers).  The write path of a CU retires roughly one distinct
      // cache line per 5-8 cycles whatever its fill, so storing straight from the MFMA layout
      // (16 rows x 32..64 B per instruction) made the epilogue cost as much as 5 K-steps.  Instead
      // each wave transposes its outputs through the LDS stage that the last K-step has just freed
      // (the other stage already holds the next tile's first slab) and writes whole rows: every
      // store instruction covers 4 (fp32) or 8 (bf16) full 256 / 128-byte row segments.
      constexpr int TS = 68;                                   // padded row stride (floats): conflict-free b128 writes
      float *tr = reinterpret_cast<float *>(lds + ((cur + NSTAGE - 1) % NSTAGE) * STAGE + wave * (STAGE / NW));   // >= 6 KiB per wave
      const int l15 = lane & 15;
      const int ncol0 = n0 + wc * TN * 16;
      // column addend (bias, or |g|^2 of the distance epilogue) for the columns this lane writes back
      const float *colv = MODE == EPI_L2DIST ? p.epi.coln : p.bias;
      float cadd[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cadd[e] = 0.f;
      if (colv) {
        if constexpr (sizeof(OutT) == 4) {
          const float4 c4 = *reinterpret_cast<const float4 *>(colv + ncol0 + l15 * 4);
          cadd[0] = c4.x; cadd[1] = c4.y; cadd[2] = c4.z; cadd[3] = c4.w;
        } else {
          const float4 c0 = *reinterpret_cast<const float4 *>(colv + ncol0 + (lane & 7) * 8);
          const float4 c1 = *reinterpret_cast<const float4 *>(colv + ncol0 + (lane & 7) * 8 + 4);
          cadd[0] = c0.x; cadd[1] = c0.y; cadd[2] = c0.z; cadd[3] = c0.w;
          cadd[4] = c1.x; cadd[5] = c1.y; cadd[6] = c1.z; cadd[7] = c1.w;
        }
      }
      auto fin = [&](float a, float add, float rnv) -> float {
        float v = MODE == EPI_L2DIST ? rnv + add - 2.0f * a : a + add;
        if (MODE == VTC_EPI_GELU) v = quick_gelu<sizeof(T) == 4>(v);
        if (MODE == EPI_SCALE) v *= scale;
        return v;
      };
      // residual mode: the x rows of pass i+1 are fetched while pass i is transposed and stored
      auto x_ptr = [&](int i, int k) -> float * {
        const int m = m0 + (wr * TM + i) * 16 + (lane >> 4) + 4 * k;
        return reinterpret_cast<float *>(p.out) + (size_t)m * ldo + ncol0 + l15 * 4;
      };
      float4 xc[4], xn[4];
      if (MODE == VTC_EPI_RESID) {
#pragma unroll
        for (int k = 0; k < 4; ++k) xc[k] = *reinterpret_cast<const float4 *>(x_ptr(0, k));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (MODE == VTC_EPI_RESID && i + 1 < TM) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xn[k] = *reinterpret_cast<const float4 *>(x_ptr(i + 1, k));
        }
        // 1. registers -> LDS: final fp32 values in [16 rows][64 cols]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          *reinterpret_cast<float4 *>(tr + l15 * TS + 16 * j + 4 * g) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 2. LDS -> global, row-contiguous
        const int mrow0 = m0 + (wr * TM + i) * 16;
        if constexpr (sizeof(OutT) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = (lane >> 4) + 4 * k, cc = l15 * 4;
            float4 v = *reinterpret_cast<const float4 *>(tr + r * TS + cc);
            const int m = mrow0 + r;
            {
              const float rnv = MODE == EPI_L2DIST ? p.epi.rown[m] : 0.f;
              v.x = fin(v.x, cadd[0], rnv); v.y = fin(v.y, cadd[1], rnv); v.z = fin(v.z, cadd[2], rnv); v.w = fin(v.w, cadd[3], rnv);
            }
            size_t orow = (size_t)m;
            bool live = true;
            if (MODE == VTC_EPI_RESID && p.epi.skip_mod > 0 && (m % p.epi.skip_mod) == 0) live = false;
            if (MODE == EPI_PATCH) {
              const int np = m % p.epi.P, ft = m / p.epi.P;
              const int tt = ft % p.epi.F, item = ft / p.epi.F;
              orow = p.epi.frames_major ? (size_t)item * p.epi.T + 1 + (size_t)tt * p.epi.P + np
                                        : (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
              const float4 p4 = *reinterpret_cast<const float4 *>(p.epi.pos + (size_t)(1 + np) * p.N + ncol0 + cc);
              v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
              if (p.epi.temporal) {
                const float4 t4 = *reinterpret_cast<const float4 *>(p.epi.temporal + (size_t)tt * p.N + ncol0 + cc);
                v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
              }
            }
            float *o = reinterpret_cast<float *>(p.out) + orow * ldo + ncol0 + cc;
            if (MODE == VTC_EPI_RESID) {
              if (live) {
                const float4 x = xc[k];
                *reinterpret_cast<float4 *>(o) = make_float4(x.x + v.x, x.y + v.y, x.z + v.z, x.w + v.w);
              }
            } else {
              *reinterpret_cast<float4 *>(o) = v;
            }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int r = (lane >> 3) + 8 * k, cc = (lane & 7) * 8;
            float4 v0 = *reinterpret_cast<const float4 *>(tr + r * TS + cc);
            float4 v1 = *reinterpret_cast<const float4 *>(tr + r * TS + cc + 4);
            v0.x = fin(v0.x, cadd[0], 0.f); v0.y = fin(v0.y, cadd[1], 0.f); v0.z = fin(v0.z, cadd[2], 0.f); v0.w = fin(v0.w, cadd[3], 0.f);
            v1.x = fin(v1.x, cadd[4], 0.f); v1.y = fin(v1.y, cadd[5], 0.f); v1.z = fin(v1.z, cadd[6], 0.f); v1.w = fin(v1.w, cadd[7], 0.f);
            uint4 pk;
            pk.x = (unsigned)f2bf(v0.x) | ((unsigned)f2bf(v0.y) << 16);
            pk.y = (unsigned)f2bf(v0.z) | ((unsigned)f2bf(v0.w) << 16);
            pk.z = (unsigned)f2bf(v1.x) | ((unsigned)f2bf(v1.y) << 16);
            pk.w = (unsigned)f2bf(v1.z) | ((unsigned)f2bf(v1.w) << 16);
            bf16_t *o = reinterpret_cast<bf16_t *>(p.out) + (size_t)(mrow0 + r) * ldo + ncol0 + cc;
            *reinterpret_cast<uint4 *>(o) = pk;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (MODE == VTC_EPI_RESID) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xc[k] = xn[k];
        }
      }
      if (has_next) __syncthreads();       // the transposition area becomes the next K-step's staging buffer
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
            orow = p.epi.frames_major ? (size_t)item * p.epi.T + 1 + (size_t)tt * p.epi.P + np
                                      : (size_t)item * p.epi.T + 1 + (size_t)np * p.epi.F + tt;
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
    li += nb_x; m0 = m0n; n0 = n0n;
    tile_offsets<AG>(a_off, m0, p.M, p.lda_bytes, wave_u, lane);
    tile_offsets<WG>(w_off, n0, p.N, p.ldw_bytes, wave_u, lane);
    has_next = li + nb_x < nt_x;
    if (has_next) decode(start_x + li + nb_x, m0n, n0n);
  }
}

}  // namespace
namespace vtcgemm {
static int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
    if (g_num_cus <= 0) g_num_cus = 256;
  }
  return g_num_cus;
}
}  // namespace vtcgemm
namespace {

template <typename T, int MODE, typename OutT, int WM, int WN, int TM, int TN, int NSTAGE>
int run(GemmParams p, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, NT_ = WM * WN * 64;
  p.MT = cdiv(p.M, BM); p.NT = cdiv(p.N, BN);
  const int ntiles = p.MT * p.NT;
  const size_t shmem = (size_t)NSTAGE * (BM + BN) * ROWB;
  const int wg_per_cu = shmem > 80 * 1024 ? 1 : 2;
  const int grid = min(ntiles, num_cus() * wg_per_cu);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<T, MODE, OutT, WM, WN, TM, TN, NSTAGE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, MODE, OutT, WM, WN, TM, TN, NSTAGE>), dim3(grid), dim3(NT_), shmem, stream, p);
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
    const int kst = p.K / Mma<T>::KPR;
    if (g_force_tile == 3 && kst >= 2) return run<T, MODE, OutT, 4, 2, 4, 4, 3>(p, stream);   // 256x128, 3-stage ring
    if (big) return run<T, MODE, OutT, 2, 4, 8, 4, 2>(p, stream);
  }
  return run<T, MODE, OutT, 2, 2, 4, 4, 2>(p, stream);
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
  { static int ea = -1; if (ea < 0) { const char *e = getenv("VTC_GEMM_EXP"); ea = e ? atoi(e) : 0; } p.exp_arg = ea; }
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
