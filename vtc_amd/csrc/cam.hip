// cam.hip -- the Context Adapter Module as ONE launch (small batches: the reference's own operating points).
//
// Replaces, for B (1 + nc) <= ~500 tokens, the ~30 generic launches of vtc_cam_forward (towers.hip) that stand for
// model/model.py:141-205 (_adapt_feature) + :207-214 (_load_comment_features' masking) + clip.model.Transformer(width = 512,
// layers = 2, heads = 8) (:396-398):
//     X = normalize(stack([main, *aux]))                        :150-151   (empty comment -> mask_embedding, :208-212)
//     per layer:  x += out_proj(MHA(ln_1 x));  x += c_proj(QuickGELU(c_fc(ln_2 x)))      (sequence = the 1 + nc tokens of an item)
//     r = normalize(mean_i normalize(Y_i))                      :157-159   (init_from_avg; residual activation :65-77)
//     adapted = normalize(normalize(main) + r)                  :203
// all in fp32 (the last op before the similarity keeps its parity margin).
//
// Why one COOPERATIVE launch and not "one workgroup per item": at these sizes the work is the weights -- 12.6 MB of fp32 per
// layer against a few hundred token rows -- and one CU pulls ~60 GB/s, so a workgroup that owns its items end to end would
// stream a layer's weights in ~200 us.  Here every phase of a layer is spread over ALL CUs by OUTPUT COLUMNS: a wave keeps a
// 16-column weight slice [16 x K] in its quarter of the workgroup's LDS and streams the token rows past it, 16 rows per fp32 MFMA
// tile (v_mfma_f32_16x16x4_f32: exact fp32; the k order inside a 16-float chunk is permuted identically for both operands), so
// a layer's weights are read ONCE, 49 KB per CU.  Phases of a layer, each closed by a grid barrier:
//     P1  ln_1 (fp32 LayerNorm of the streamed row tile) + in_proj + bias                        -> qkv   [rows, 3 D]
//         (layer 0 builds the normalised tokens from main / comments / mask_embedding on the way and stores x)
//     P2  softmax(q k^T / 8) v per (item, head): one wave each, lane = head dimension              -> att   [rows, D]
//     P3  out_proj + bias + residual                                                               -> x
//     P4  ln_2 + c_fc + bias + QuickGELU                                                           -> hid   [rows, 4 D]
//     P5  c_proj (K = 4 D: the four waves of a workgroup take a quarter each, LDS reduction) + bias + residual -> x
// then the finalisation per item (one wave).  Hand-off between phases WITHOUT fences (MI355X_MICROARCH "Valid forms", first row
// of its table; cdna_hip_programming Guideline 16 R1): every byte one workgroup hands to another -- x, qkv, the attention output,
// the MLP hidden -- is stored write-through (`sc1`, 4- or 16-byte buffer stores) and read with `sc1` loads ONLY (they bypass
// the CU's L1, the one cache that can hold a stale copy); at a barrier every storing wave drains (vmcnt(0)), the workgroup
// meets, ONE lane adds to a monotonic agent-scope counter and polls it with sc1 loads + s_sleep, and the workgroup meets again
// before anybody loads.  (The fence form -- release + acquire per barrier -- measured 20 us per phase.)  The spin is BOUNDED:
// a barrier that does not complete sets an error word and lets the grid drain (garbage results, never a hung GPU).
// 10 barriers + the phases: see DESIGN.md for the measured times against the multi-launch path (which stays for larger batches:
// there the GEMMs fill the chip and column-splitting would re-read the activations once per 16 columns).
#include "common.h"

namespace {

constexpr int CAM_MAX_LAYERS = 4;
constexpr size_t CAM_BAR_BYTES = 32 * 17 * 4;       // grid_barrier: top + error, 8 group counters, 8 generation words

struct CamFusedParams {
  const float *main_f, *comm, *mask_emb;
  const int64_t *comments;
  int B, nc, ctx, Lc, rows, ntiles, layers, heads;
  // folded LayerNorms (vtc_block_w qkv_wf / qkv_s / qkv_c, fc_wf / fc_s / fc_c, fp32) + the two plain projections
  const float *qkv_wf[CAM_MAX_LAYERS], *qkv_s[CAM_MAX_LAYERS], *qkv_c[CAM_MAX_LAYERS], *out_w[CAM_MAX_LAYERS], *out_b[CAM_MAX_LAYERS];
  const float *fc_wf[CAM_MAX_LAYERS], *fc_s[CAM_MAX_LAYERS], *fc_c[CAM_MAX_LAYERS], *proj_w[CAM_MAX_LAYERS], *proj_b[CAM_MAX_LAYERS];
  float *x, *big, *att;       // residual stream [rows, D]; qkv [rows, 3 D] / MLP hidden [rows, 4 D] (never live together); attention output
  int *bar;                   // grid-barrier words (grid_barrier; CAM_BAR_BYTES, zeroed before the launch); bar[1]: error word
  int *err_host;              // device-visible pinned host word (never reset): a barrier that gave up sets it, the next launch reports it
  unsigned long long *stamps; // diagnostics (VTC_CAM_STAMPS=1): s_memrealtime of workgroup 0 at the start and after every barrier; else NULL
  int act;
  float scale;
  const float *bn_mean, *bn_var;
  float *out;                 // adapted [B, D]
};

// write-through / L1-bypassing accesses to the buffers the workgroups hand to one another (aux 16 = sc1)
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
struct Sc1Buf {
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ Sc1Buf(const void *base, size_t bytes) {
    r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(bytes < 0xFFFFFFF0u ? bytes : 0xFFFFFFF0u), 0x00020000);
  }
  __device__ __forceinline__ float4 ld16(int off) const {
    const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  __device__ __forceinline__ void st16(int off, float4 v) const {
    const v4u_t vv = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(vv, r, off, 0, 16);
  }
  __device__ __forceinline__ float ld4(int off) const { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 16)); }
  __device__ __forceinline__ void st4(int off, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 16); }
};

// sum over the four lanes (g = 0..3) that share a row
__device__ __forceinline__ float row_sum4(float s) {
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// padded row stride (floats) of a weight-slice image in LDS
template <int D> constexpr int WLD = D + 4;
// rows [n0, n0 + 16) x columns [k0, k0 + D) of W (leading dimension ld) -> this wave's LDS image; lane (g, m) moves W[n0 + m][k0 + 16 q + 4 g ..]
// to exactly the place it will read back, so only the wave's own program order is involved (no barrier)
template <int D>
__device__ __forceinline__ void stage_slice(float *wl, const float *__restrict__ W, size_t ld, int n0, int k0, int lane) {
  const float *src = W + (size_t)(n0 + (lane & 15)) * ld + k0 + 4 * (lane >> 4);
  float4 v[D / 16];                 // every load in flight before the first LDS write: one memory latency per slice, not four
#pragma unroll
  for (int q = 0; q < D / 16; ++q) v[q] = *reinterpret_cast<const float4 *>(src + 16 * q);
#pragma unroll
  for (int q = 0; q < D / 16; ++q) *reinterpret_cast<float4 *>(wl + 16 * q) = v[q];
}

constexpr long long CAM_SPIN_LIMIT = 400000;      // polls (each ~1 us with its s_sleep): a barrier that takes > ~0.4 s is declared dead

// Grid barrier number `phase` (1, 2, ...), two levels so that nobody's arrival or poll shares a cache line with more than 31
// others: workgroups with equal (blockIdx % 8) -- one XCD under the observed round-robin placement; any fixed grouping is
// correct -- add to their group's counter; the group's last arriver adds to the top counter, waits for all groups there and
// publishes the phase number in the group's generation word, which the rest of the group polls.  Counters are monotonic over
// the launch (zeroed before it); every word sits on a 128-byte line of its own.  No fences: hand-off rule at the top.
//   bar[0] top, bar[1] error word, bar[32 (1 + g)] group g's arrivals, bar[32 (9 + g)] group g's generation
__device__ __forceinline__ void grid_barrier(int *bar, int phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores ...
  __syncthreads();                                        // ... before the one lane that signals for all of them
  if (threadIdx.x == 0) {
    const int ngroups = min(8, (int)gridDim.x), grp = blockIdx.x % ngroups;
    const int gsize = ((int)gridDim.x + ngroups - 1 - grp) / ngroups;
    int *cnt = bar + 32 * (1 + grp), *gen = bar + 32 * (9 + grp);
    long long spins = 0;
    const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == phase * gsize) {
      __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * ngroups && ++spins <= CAM_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(gen, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase && ++spins <= CAM_SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
    }
    if (spins > CAM_SPIN_LIMIT) __hip_atomic_store(bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // a barrier did not complete: the finalisation writes NaN
  }
  __syncthreads();                                        // nobody loads before the poll has matched
}

__device__ __forceinline__ float act_apply_cam(int act, float v, float msq, float scale) {
  // model/model.py:34-39,65-77 (as embed.hip act_apply)
  if (act == VTC_ACT_NORMALIZE) return (v + 1e-9f) / sqrtf(msq);
  if (act == VTC_ACT_SQUASH) {
    const float mag = sqrtf(msq);
    return scale * (msq / (1.0f + msq)) * ((v + 1e-9f) / mag);
  }
  if (act == VTC_ACT_TANH) return tanhf(v);
  return v;
}

// work split of a projection phase with N output columns: `wps` waves share a 16-column slice and split its row tiles
__device__ __forceinline__ int waves_per_slice(const CamFusedParams &p, int N) {
  return max(1, min(p.ntiles, (int)(gridDim.x * 4) / (N / 16)));
}

// The slice of this wave's FIRST unit of a projection phase, staged ahead of time (weights do not depend on any barrier: the
// load rides under the wait for the phase in front)
template <int D>
__device__ __forceinline__ void prestage_proj(float *wlds, const CamFusedParams &p, const float *__restrict__ W, int N) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int wps = waves_per_slice(p, N);
  if (gw < (N / 16) * wps)
    stage_slice<D>(wlds + (threadIdx.x >> 6) * (16 * WLD<D>) + (lane & 15) * WLD<D> + 4 * (lane >> 4), W, D, 16 * (gw / wps), 0, lane);
}

// A projection phase: out[r][n0 + c] = epi( sum_k W[n0 + c][k] * in(r)[k] + bias ), K = D, N output columns, over all row tiles.
//   MODE 0: plain store (ld = N)     1: QuickGELU store     2: residual (x += ...; N == D)
//   LN: the LayerNorm in front is FOLDED as in the towers (include/vtc_hip.h vtc_block_w *_wf / *_s / *_c, here in fp32):
//       LN(x) W^T + b = rstd (x W'^T - mean s) + c,  W' = gamma . W,  s = row sums of W',  c = b + W beta -- the raw row streams
//       past the slice ONCE, the MFMAs and the row statistics (shifted sums: the shift is the row's first element, so nothing
//       cancels beyond a few ulp) run on the same pass (a LayerNorm applied first needs the tile twice, and the passes are
//       latency chains: measured 6 us per tile against 2.5);
//   TOK: the input rows are the CAM tokens X = normalize(src), src = main / comment embedding / mask_embedding (layer 0's P1).
//       LN(src / |src|) = (src - mean) / sqrt(var + eps |src|^2) on the statistics of src itself: still one pass; the waves of
//       slice 0 then store X = src / |src| to x.
// The wave's weight slice [16 x D] sits in ITS quarter of the workgroup's LDS (rows padded by 4 floats) -- in registers it took
// D / 4 VGPRs next to the row tile and the kernel spilled 700-900 registers (measured).  The lane's share of a row tile (D / 4
// VGPRs) is loaded in one go: the phase is a chain of memory latencies, not of bytes.
template <int D, int MODE, bool LN, bool TOK>
__device__ __forceinline__ void proj_phase(float *wlds, const CamFusedParams &p, const float *__restrict__ in, const float *__restrict__ W, const float *__restrict__ bias,
                                           const float *__restrict__ fs, float *__restrict__ out, int N, bool prestaged) {
  constexpr int CH = 8, NP = D / 16 / CH;        // chunks of 16 floats per piece, pieces per row
  static_assert(D % (16 * CH) == 0, "row = whole pieces");
  const int lane = threadIdx.x & 63;
  const int TW = gridDim.x * 4, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nslices = N / 16;
  const int wps = waves_per_slice(p, N);
  const int g = lane >> 4, n = lane & 15;
  const Sc1Buf bin(TOK ? (const void *)p.x : (const void *)in, (size_t)p.rows * D * 4), bout(out, (size_t)p.rows * N * 4), bx(p.x, (size_t)p.rows * D * 4);
  float *wl = wlds + (threadIdx.x >> 6) * (16 * WLD<D>) + n * WLD<D> + 4 * g;     // this lane's place in the wave's slice image
  for (int u = gw; u < nslices * wps; u += TW) {
    const int slice = u / wps, sub = u - slice * wps;
    if (!(prestaged && u == gw)) stage_slice<D>(wl, W, D, 16 * slice, 0, lane);
    const float4 b4 = *reinterpret_cast<const float4 *>(bias + 16 * slice + 4 * g);       // LN: c = b + W beta
    [[maybe_unused]] float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LN) s4 = *reinterpret_cast<const float4 *>(fs + 16 * slice + 4 * g);
    for (int t = sub; t < p.ntiles; t += wps) {
      const int r = min(16 * t + n, p.rows - 1);
      const float *src = nullptr;
      [[maybe_unused]] const int soff = (r * D + 4 * g) * 4;      // byte offset of the lane's first chunk in a [rows, D] buffer
      if constexpr (TOK) {
        // X[b Lc + 0] = normalize(main[b]);  X[b Lc + 1 + c] = normalize(empty(b, c) ? mask_embedding : comm[b nc + c])
        const int b = r / p.Lc, tk = r - b * p.Lc;
        if (tk == 0) src = p.main_f + (size_t)b * D;
        else {
          const int ci = b * p.nc + (tk - 1);
          src = p.comments[(size_t)ci * p.ctx + 1] == 49407 ? p.mask_emb : p.comm + (size_t)ci * D;      // model/model.py:208
        }
        src += 4 * g;
      }
      auto piece = [&](int pc, int c) -> float4 {
        if constexpr (TOK) return *reinterpret_cast<const float4 *>(src + 16 * (pc * CH + c));
        else return bin.ld16(soff + 64 * (pc * CH + c));
      };
      // four independent accumulator chains (one per component of a chunk): a single chain of D / 4 dependent fp32 MFMAs is
      // ~36 cycles of latency each (2.4 us at D = 512) -- a visible share of a phase that is a chain of latencies
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
      float4 xt[D / 16];            // the lane's share of the row tile, every load in flight at once (one latency per tile)
#pragma unroll
      for (int q = 0; q < D / 16; ++q) xt[q] = piece(q / CH, q % CH);
      [[maybe_unused]] const float x0 = __shfl(xt[0].x, n, 64);      // the row's first element (its g = 0 lane holds it)
      [[maybe_unused]] float s1 = 0.f, s2 = 0.f, sq = 0.f;
#pragma unroll
      for (int q = 0; q < D / 16; ++q) {
        const float4 v = xt[q];
        if constexpr (LN) {
          const float a = v.x - x0, b = v.y - x0, cc = v.z - x0, d = v.w - x0;
          s1 += (a + b) + (cc + d);
          s2 += (a * a + b * b) + (cc * cc + d * d);
        }
        if constexpr (TOK) sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        const float4 wq = *reinterpret_cast<const float4 *>(wl + 16 * q);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.x, v.x, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.y, v.y, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.z, v.z, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.w, v.w, acc3, 0, 0, 0);
      }
      acc = (acc + acc1) + (acc2 + acc3);
      float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
      [[maybe_unused]] float nrm = 1.0f;
      if constexpr (LN) {
        s1 = row_sum4(s1); s2 = row_sum4(s2);
        const float dm = s1 * (1.0f / D), mean = x0 + dm;
        float eps = 1e-5f;
        if constexpr (TOK) {
          const float n2 = row_sum4(sq);
          nrm = sqrtf(n2);
          eps *= n2;                                        // LN(src / |src|): the eps of the normalised row, in src's units
        }
        const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / D) - dm * dm, 0.f) + eps);
        v.x = rstd * (v.x - mean * s4.x); v.y = rstd * (v.y - mean * s4.y); v.z = rstd * (v.z - mean * s4.z); v.w = rstd * (v.w - mean * s4.w);
      }
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      const int rr = 16 * t + n;
      if (rr < p.rows) {
        const int ooff = (rr * N + 16 * slice + 4 * g) * 4;
        if constexpr (MODE == 1) {
          v.x = v.x / (1.0f + expf(-1.702f * v.x)); v.y = v.y / (1.0f + expf(-1.702f * v.y));
          v.z = v.z / (1.0f + expf(-1.702f * v.z)); v.w = v.w / (1.0f + expf(-1.702f * v.w));
        }
        if constexpr (MODE == 2) {
          const float4 xo = bout.ld16(ooff);
          v.x += xo.x; v.y += xo.y; v.z += xo.z; v.w += xo.w;
        }
        bout.st16(ooff, v);
        if constexpr (TOK) {
          if (slice == 0) {       // X = src / |src| (division, as the reference's x / x.norm()) for the residual stream
#pragma unroll
            for (int q = 0; q < D / 16; ++q) {
              float4 xv = xt[q];
              xv.x /= nrm; xv.y /= nrm; xv.z /= nrm; xv.w /= nrm;
              bx.st16(soff + 64 * q, xv);
            }
          }
        }
      }
    }
  }
}

template <int D>
__global__ __launch_bounds__(256, 1) void cam_fused_kernel(const CamFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];       // 4 weight-slice images [16][D + 4], then a scratch area
  float *scratch = wlds + 4 * 16 * WLD<D>;                          // 4 x 2 x 15 x 64 floats: P2's q / k rows per wave; P5's reduction buffer
  float(*red)[64][4] = reinterpret_cast<float(*)[64][4]>(scratch);
  float sc_hi[3] = {0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TW = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  const int g = lane >> 4, n = lane & 15;
  const Sc1Buf bx(p.x, (size_t)p.rows * D * 4), bbig(p.big, (size_t)p.rows * 4 * D * 4), batt(p.att, (size_t)p.rows * D * 4);
  if (p.stamps && threadIdx.x == 0 && blockIdx.x == 0) p.stamps[0] = __builtin_amdgcn_s_memrealtime();
  int phase = 0;
  // diagnostics (VTC_CAM_STAMPS): workgroup 0's clock after each phase's compute (slot 3 i + 1), after the prestage of the next
  // weights (3 i + 2) and after the barrier (3 i + 3)
  auto stamp = [&](int slot) {
    if (p.stamps && threadIdx.x == 0 && blockIdx.x == 0) p.stamps[slot] = __builtin_amdgcn_s_memrealtime();
  };
  auto sync_grid = [&]() {
    stamp(3 * phase + 2);
    ++phase;
    grid_barrier(p.bar, phase);
    stamp(3 * phase);
  };
  // P5's first unit of this workgroup: slice (-1: none), staged ahead like the projection phases'
  const int p5_slices = D / 16, p5_wgps = max(1, min(p.ntiles, (int)gridDim.x / p5_slices));
  auto prestage_p5 = [&](int l) {
    if ((int)blockIdx.x < p5_slices * p5_wgps)
      stage_slice<D>(wlds + wave * (16 * WLD<D>) + n * WLD<D> + 4 * g, p.proj_w[l], 4 * D, 16 * ((int)blockIdx.x / p5_wgps), wave * D, lane);
  };
  prestage_proj<D>(wlds, p, p.qkv_wf[0], 3 * D);
  for (int l = 0; l < p.layers; ++l) {
    // ---- P1: ln_1 (folded) + in_proj -> qkv (layer 0: from the tokens) -----------------------------------------------------
    if (l == 0) proj_phase<D, 0, true, true>(wlds, p, nullptr, p.qkv_wf[l], p.qkv_c[l], p.qkv_s[l], p.big, 3 * D, true);
    else proj_phase<D, 0, true, false>(wlds, p, p.x, p.qkv_wf[l], p.qkv_c[l], p.qkv_s[l], p.big, 3 * D, true);
    stamp(3 * phase + 1);
    sync_grid();
    // ---- P2: attention core, one wave per (item, head) ---------------------------------------------------------------------
    // lane = head dimension d for the loads and the output; the Lc x Lc scores are taken by one lane each from q / k rows staged in
    // the wave's LDS scratch (16 float4 reads + 64 FMA per score; as wave-wide reductions they were 36 shuffle chains: 19 us).
    // Waves without an attention unit stage P3's weight slice meanwhile; the others do it afterwards.
    {
      const bool has_unit = gw < p.B * p.heads;
      if (!has_unit) prestage_proj<D>(wlds, p, p.out_w[l], D);
      float *qs = scratch + wave * (2 * 15 * 64), *ks = qs + 15 * 64;       // [Lc][64] each (Lc <= 15), then the scores over qs
      for (int u = gw; u < p.B * p.heads; u += TW) {
        const int b = u / p.heads, h = u - b * p.heads;
        const int base = (b * p.Lc * 3 * D + h * 64 + lane) * 4;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < p.Lc) {
            qs[i * 64 + lane] = bbig.ld4(base + i * 3 * D * 4) * 0.125f;       // q * head_dim^-0.5 (timesformer_clip_alt.py:52 / upstream MHA)
            ks[i * 64 + lane] = bbig.ld4(base + (i * 3 * D + D) * 4);
            v[i] = bbig.ld4(base + (i * 3 * D + 2 * D) * 4);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float sc = 0.f;
        for (int e = lane; e < p.Lc * p.Lc; e += 64) {       // (Lc <= 8: one pass; Lc <= 15: up to four)
          const int i = e / p.Lc, j = e - i * p.Lc;
          float s = 0.f;
#pragma unroll
          for (int d4 = 0; d4 < 16; ++d4) {
            const float4 a = *reinterpret_cast<const float4 *>(qs + i * 64 + 4 * d4), c = *reinterpret_cast<const float4 *>(ks + j * 64 + 4 * d4);
            s += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
          }
          if (e < 64) sc = s;
          else sc_hi[(e >> 6) - 1] = s;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();           // every lane is past its q / k reads: the scores may overwrite qs
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < p.Lc * p.Lc) qs[lane] = sc;
        for (int e = lane + 64; e < p.Lc * p.Lc; e += 64) qs[e] = sc_hi[(e >> 6) - 1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < p.Lc) {
            float mx = -INFINITY;
            for (int j = 0; j < p.Lc; ++j) mx = fmaxf(mx, qs[i * p.Lc + j]);
            float den = 0.f, o = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
              if (j < p.Lc) {
                const float ex = expf(qs[i * p.Lc + j] - mx);
                den += ex;
                o += ex * v[j];
              }
            batt.st4(((b * p.Lc + i) * D + h * 64 + lane) * 4, o / den);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();           // before the next unit overwrites the scratch
      }
      if (has_unit) prestage_proj<D>(wlds, p, p.out_w[l], D);
    }
    stamp(3 * phase + 1);
    sync_grid();
    // ---- P3: out_proj + residual ---------------------------------------------------------------------------------------------
    proj_phase<D, 2, false, false>(wlds, p, p.att, p.out_w[l], p.out_b[l], nullptr, p.x, D, true);
    stamp(3 * phase + 1);
    prestage_proj<D>(wlds, p, p.fc_wf[l], 4 * D);
    sync_grid();
    // ---- P4: ln_2 (folded) + c_fc + QuickGELU -> hidden ------------------------------------------------------------------------
    proj_phase<D, 1, true, false>(wlds, p, p.x, p.fc_wf[l], p.fc_c[l], p.fc_s[l], p.big, 4 * D, true);
    stamp(3 * phase + 1);
    prestage_p5(l);
    sync_grid();
    // ---- P5: c_proj (K = 4 D, a quarter per wave of the workgroup) + residual ---------------------------------------------------
    {
      const int nslices = p5_slices, wgps = p5_wgps;
      for (int u = blockIdx.x; u < nslices * wgps; u += gridDim.x) {
        const int slice = u / wgps, sub = u - slice * wgps;
        float *wl = wlds + wave * (16 * WLD<D>) + n * WLD<D> + 4 * g;
        if (u != (int)blockIdx.x) stage_slice<D>(wl, p.proj_w[l], 4 * D, 16 * slice, wave * D, lane);
        const float4 b4 = *reinterpret_cast<const float4 *>(p.proj_b[l] + 16 * slice + 4 * g);
        for (int t = sub; t < p.ntiles; t += wgps) {
          const int soff = (min(16 * t + n, p.rows - 1) * 4 * D + wave * D + 4 * g) * 4;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          float4 xv[D / 16];
#pragma unroll
          for (int q = 0; q < D / 16; ++q) xv[q] = bbig.ld16(soff + 64 * q);
#pragma unroll
          for (int q = 0; q < D / 16; ++q) {
            const float4 wq = *reinterpret_cast<const float4 *>(wl + 16 * q);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.x, xv[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.y, xv[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.z, xv[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.w, xv[q].w, acc, 0, 0, 0);
          }
          *reinterpret_cast<float4 *>(&red[wave][lane][0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          __syncthreads();
          if (wave == 0) {
            const int r = 16 * t + n;
            if (r < p.rows) {
              const int ooff = (r * D + 16 * slice + 4 * g) * 4;
              const float4 a0 = *reinterpret_cast<const float4 *>(&red[0][lane][0]), a1 = *reinterpret_cast<const float4 *>(&red[1][lane][0]);
              const float4 a2 = *reinterpret_cast<const float4 *>(&red[2][lane][0]), a3 = *reinterpret_cast<const float4 *>(&red[3][lane][0]);
              const float4 xo = bx.ld16(ooff);
              bx.st16(ooff, make_float4(xo.x + (((a0.x + a1.x) + (a2.x + a3.x)) + b4.x), xo.y + (((a0.y + a1.y) + (a2.y + a3.y)) + b4.y),
                                                           xo.z + (((a0.z + a1.z) + (a2.z + a3.z)) + b4.z), xo.w + (((a0.w + a1.w) + (a2.w + a3.w)) + b4.w)));
            }
          }
          __syncthreads();
        }
      }
    }
    stamp(3 * phase + 1);
    if (l + 1 < p.layers) prestage_proj<D>(wlds, p, p.qkv_wf[l + 1], 3 * D);
    sync_grid();
  }
  // ---- finalisation, one wave per item (as embed.hip cam_finalize_kernel, init_from_avg):
  // r = normalize(mean_i normalize(Y_i)); r = act(r); adapted = normalize(normalize(main) + r)       model/model.py:157-159,65-77,203
  constexpr int MD = D / 64;
  // A grid barrier that gave up (bar[1], CAM_SPIN_LIMIT: this launch did not have the chip to itself for ~0.4 s) means rows of x
  // may be stale.  Fail LOUDLY: the items this wave owns come out as NaN, which no similarity, loss or R@K survives quietly.
  const bool dead = __hip_atomic_load(p.bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (dead && p.err_host && threadIdx.x == 0) __hip_atomic_store(p.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int b = gw; b < p.B; b += TW) {
    if (dead) {
#pragma unroll
      for (int k = 0; k < MD; ++k) p.out[(size_t)b * D + lane + 64 * k] = __uint_as_float(0x7FC00000u);
      continue;
    }
    float r[MD];
#pragma unroll
    for (int k = 0; k < MD; ++k) r[k] = 0.f;
    for (int t = 0; t < p.Lc; ++t) {
      const int yoff = ((b * p.Lc + t) * D + lane) * 4;
      float v[MD], s = 0.f;
#pragma unroll
      for (int k = 0; k < MD; ++k) {
        v[k] = bx.ld4(yoff + 256 * k);
        s += v[k] * v[k];
      }
      const float nrm = sqrtf(wave_sum(s));
#pragma unroll
      for (int k = 0; k < MD; ++k) r[k] += v[k] / nrm;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MD; ++k) { r[k] /= p.Lc; s += r[k] * r[k]; }
    const float nrm = sqrtf(wave_sum(s));
#pragma unroll
    for (int k = 0; k < MD; ++k) r[k] /= nrm;
    if (p.act == VTC_ACT_SUB_MEAN || p.act == VTC_ACT_BN) {
#pragma unroll
      for (int k = 0; k < MD; ++k) {
        const int c = lane + 64 * k;
        const float d = r[k] - p.bn_mean[c];
        r[k] = p.act == VTC_ACT_BN ? d / sqrtf(p.bn_var[c] + 1e-5f) : d;
      }
    } else if (p.act != VTC_ACT_NONE) {
      float s2 = 0.f;
#pragma unroll
      for (int k = 0; k < MD; ++k) s2 += (r[k] + 1e-9f) * (r[k] + 1e-9f);
      const float msq = wave_sum(s2);
#pragma unroll
      for (int k = 0; k < MD; ++k) r[k] = act_apply_cam(p.act, r[k], msq, p.scale);
    }
    float m[MD], sm = 0.f;
#pragma unroll
    for (int k = 0; k < MD; ++k) {
      m[k] = p.main_f[(size_t)b * D + lane + 64 * k];
      sm += m[k] * m[k];
    }
    const float mn = sqrtf(wave_sum(sm));
    float s3 = 0.f;
#pragma unroll
    for (int k = 0; k < MD; ++k) {
      m[k] = m[k] / mn + r[k];
      s3 += m[k] * m[k];
    }
    const float an = sqrtf(wave_sum(s3));
#pragma unroll
    for (int k = 0; k < MD; ++k) p.out[(size_t)b * D + lane + 64 * k] = m[k] / an;
  }
}

}  // namespace

namespace vtcgemm { int num_cus(); }

// Largest token count B (1 + nc) the one-launch CAM takes (env VTC_CAM_FUSED_MAX_ROWS; 0 disables it): above it the GEMM launches
// of towers.hip fill the chip and win (measured crossover).
int cam_fused_max_rows() {
  static const int v = [] { const char *e = getenv("VTC_CAM_FUSED_MAX_ROWS"); return e ? atoi(e) : 512; }();
  return v;
}

bool cam_fused_supported(const vtc_cam_w *w, int B, int nc, int dtype) {
  const int Lc = 1 + nc;
  for (int l = 0; l < w->layers && l < CAM_MAX_LAYERS; ++l)      // the folded-LayerNorm weights (fp32) must have been packed
    if (!w->blocks[l].qkv_wf || !w->blocks[l].qkv_s || !w->blocks[l].qkv_c || !w->blocks[l].fc_wf || !w->blocks[l].fc_s || !w->blocks[l].fc_c) return false;
  return dtype == VTC_F32 && w->init_from_avg && !(w->flags & VTC_CAM_NO_FUSED) && (w->width == 512 || w->width == 128) &&
         w->width == w->heads * 64 && w->layers >= 1 && w->layers <= CAM_MAX_LAYERS && Lc <= 15 && B * Lc <= cam_fused_max_rows();
}

// Two cooperative launches on ONE device at the same time (two streams, two host threads) would split the CUs between them and
// each would wait at its first barrier for workgroups that cannot become resident (a width-512 workgroup takes a whole CU's
// LDS): the bounded spin would end it after seconds, with garbage.  So the launcher remembers, per device, the stream and a
// completion event of the last one-launch CAM: same stream -> ordered behind it; another stream and the event not yet complete
// -> this call reports "busy" and vtc_cam_forward takes the multi-launch path (same results).  No synchronisation.
#include <mutex>
namespace {
int *g_cam_err_words[64] = {};     // per device: pinned, device-visible; set by a grid barrier that gave up, never reset
struct CamInFlight { hipStream_t stream = nullptr; hipEvent_t done = nullptr; bool any = false; };
CamInFlight g_cam_inflight[64];
std::mutex g_cam_mu;
}  // namespace
// (check, launch and mark happen under one lock: see launch_cam_fused)
static bool cam_fused_busy_locked(int dev, hipStream_t stream) {
  CamInFlight &f = g_cam_inflight[dev];
  if (!f.any || f.stream == stream) return false;
  return hipEventQuery(f.done) != hipSuccess;
}
static void cam_fused_mark_locked(int dev, hipStream_t stream) {
  CamInFlight &f = g_cam_inflight[dev];
  if (!f.done && hipEventCreateWithFlags(&f.done, hipEventDisableTiming) != hipSuccess) { f.done = nullptr; f.any = false; return; }
  f.stream = stream; f.any = true;
  (void)hipEventRecord(f.done, stream);
}

// x [rows, D], big [rows, 4 D], att [rows, D] fp32; bar: CAM_BAR_BYTES (17 words on 128-byte lines of their own).
// Returns 0 on success, 1 on error, -1 when another stream's one-launch CAM may still be running on this device (nothing was
// enqueued: the caller takes the multi-launch path).
size_t cam_fused_bar_bytes() { return CAM_BAR_BYTES; }
int launch_cam_fused(const vtc_cam_w *w, const float *main_feats, const float *comm_feats, const int64_t *comments, int ctx, int B, int nc,
                     float *adapted, float *x, float *big, float *att, int *bar, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  std::lock_guard<std::mutex> lk(g_cam_mu);
  // the error word of this device: pinned, device-visible, allocated once.  A grid barrier that gave up (the launch lost its
  // residency guarantee: it cannot under a cooperative launch) wrote NaN embeddings AND set this word.
  if (!g_cam_err_words[dev]) {
    int *h = nullptr;
    if (hipHostMalloc((void **)&h, 64, hipHostMallocMapped) != hipSuccess || !h) return -1;     // no error word: take the multi-launch path
    *h = 0;
#ifdef VTC_TEST_HOOKS
    // test hook, compiled into vtc_amd/lib/libvtc_hip_testhooks.so ONLY (Makefile `testhooks`; tests/test_gpu_cam_fallback.py loads that build
    // in a child process): behave as if an earlier launch's barrier had given up.  The product library has no such switch.
    if (const char *e = getenv("VTC_CAM_TEST_GAVE_UP")) { if (e[0] == '1') *h = 1; }
#endif
    g_cam_err_words[dev] = h;
  }
  int **err_words = g_cam_err_words;
  // a barrier that gave up (a co-resident kernel this process cannot see held CUs: another stream's persistent
  // GEMM, RCCL, another process on the card) is a property of the MOMENT, not of the device -- so it must not fail every later
  // call.  The call it hit wrote NaN embeddings (loud) and set the word; from then on this device takes the multi-launch path
  // (same results, no grid barrier), said once on stderr; vtc_cam_fused_gave_up() lets the host ask at its next synchronisation.
  if (*(volatile int *)err_words[dev] != 0) {
    static bool told[64] = {};
    if (!told[dev]) {
      told[dev] = true;
      fprintf(stderr, "[vtc_amd] cam_fused: an earlier one-launch CAM on device %d gave up at a grid barrier (that call's embeddings were "
                      "written as NaN: vtc_cam_fused_gave_up(%d) != 0); the one-launch path is now OFF for this device, the multi-launch "
                      "path takes over\n", dev, dev);
    }
    return -1;
  }
  if (cam_fused_busy_locked(dev, stream)) return -1;
  CamFusedParams p;
  {
    void *dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, err_words[dev], 0) != hipSuccess) return -1;
    p.err_host = (int *)dptr;
  }
  p.main_f = main_feats; p.comm = comm_feats; p.mask_emb = w->mask_embedding; p.comments = comments;
  p.B = B; p.nc = nc; p.ctx = ctx; p.Lc = 1 + nc; p.rows = B * (1 + nc); p.ntiles = cdiv(p.rows, 16); p.layers = w->layers; p.heads = w->heads;
  for (int l = 0; l < w->layers; ++l) {
    const vtc_block_w &b = w->blocks[l];
    p.qkv_wf[l] = (const float *)b.qkv_wf; p.qkv_s[l] = b.qkv_s; p.qkv_c[l] = b.qkv_c;
    p.out_w[l] = (const float *)b.out_w; p.out_b[l] = b.out_b;
    p.fc_wf[l] = (const float *)b.fc_wf; p.fc_s[l] = b.fc_s; p.fc_c[l] = b.fc_c;
    p.proj_w[l] = (const float *)b.proj_w; p.proj_b[l] = b.proj_b;
  }
  p.x = x; p.big = big; p.att = att; p.bar = bar;
  static const bool want_stamps = [] { const char *e = getenv("VTC_CAM_STAMPS"); return e && e[0] == '1'; }();
  static unsigned long long *stamps = nullptr;       // diagnostics only: a buffer of its own that no kernel reads
  if (want_stamps && !stamps) (void)hipMalloc(&stamps, 64 * sizeof(unsigned long long));
  if (want_stamps && stamps) (void)hipMemsetAsync(stamps, 0, 64 * sizeof(unsigned long long), stream);
  p.stamps = want_stamps ? stamps : nullptr;
  p.act = w->residual_activation; p.scale = w->squash_scale; p.bn_mean = w->bn_mean; p.bn_var = w->bn_var;
  p.out = adapted;
  VTC_CHECK(hipMemsetAsync(bar, 0, CAM_BAR_BYTES, stream) == hipSuccess, "cam_fused: barrier reset failed");
  // Every workgroup must be resident for the grid barrier: one per CU (width 512: 136 KiB of LDS each, so exactly one fits).
  // nothing guaranteed that.  Two launch forms, measured in round 4 (profiles/r04_experiments.txt):
  //   * default -- an ordinary launch behind an occupancy check of THIS kernel on THIS device (grid <= CUs x resident workgroups per
  //     CU, else -1: the caller takes the multi-launch path).  What the check cannot see (a foreign kernel holding a CU: another
  //     process, RCCL, a CU mask) ends in the bounded spin of grid_barrier: NaN embeddings AND the device-visible error word
  //     below, which fails every later call on the device with a message -- loud, never a silent wrong answer;
  //   * VTC_CAM_COOP=1 (read once) -- hipLaunchCooperativeKernel: the runtime validates residency and serialises cooperative
  //     grids.  Not the default because it costs +0.11 ms per forward at B = 1 (+6 %; the cooperative queue hand-over) and because
  //     rocprofv3 of ROCm 7.2 segfaults in its teardown in any process that made one (tools/exit_probe.py).
  const int grid = vtcgemm::num_cus();
  ProfScope prof(VTC_PROF_GEMM_F32, 2.0 * p.rows * 12.0 * w->width * w->width * w->layers, stream);
  void *kargs[] = {&p};
  hipError_t le;
  static const bool coop = [] { const char *e = getenv("VTC_CAM_COOP"); return e && e[0] == '1'; }();
  auto launch = [&](const void *fn, int shmem) -> hipError_t {
    if (coop) return hipLaunchCooperativeKernel(fn, dim3(grid), dim3(256), kargs, shmem, stream);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, shmem) != hipSuccess || per_cu < 1) return hipErrorCooperativeLaunchTooLarge;
    return hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, shmem, stream);
  };
  vtc_count_launch();
  if (w->width == 512) {
    constexpr int shmem = (4 * 16 * WLD<512> + 4 * 2 * 15 * 64) * 4;
    static PerDeviceOnce attr;
    if (ensure_dynamic_lds(attr, reinterpret_cast<const void *>(&cam_fused_kernel<512>), shmem, "cam_fused")) return 1;
    le = launch(reinterpret_cast<const void *>(&cam_fused_kernel<512>), shmem);
  } else {
    constexpr int shmem = (4 * 16 * WLD<128> + 4 * 2 * 15 * 64) * 4;
    le = launch(reinterpret_cast<const void *>(&cam_fused_kernel<128>), shmem);
  }
  if (le != hipSuccess) {
    (void)hipGetLastError();      // (the profiler then holds one empty record for this call)
    return -1;
  }
  cam_fused_mark_locked(dev, stream);
  if (want_stamps && stamps) {     // diagnostics: synchronises
    unsigned long long h[64];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    const int nph = 5 * w->layers;
    fprintf(stderr, "[cam stamps] B=%d rows=%d: total %.1f us; per phase compute/prestage/barrier (us):", B, p.rows, (h[3 * nph] - h[0]) * 0.01);
    for (int i = 0; i < nph; ++i)
      fprintf(stderr, " %.1f/%.1f/%.1f", (h[3 * i + 1] - h[3 * i]) * 0.01, (h[3 * i + 2] - h[3 * i + 1]) * 0.01, (h[3 * i + 3] - h[3 * i + 2]) * 0.01);
    fprintf(stderr, "\n");
  }
  return 0;
}

// 1 when a one-launch CAM on `device` has ever given up at a grid barrier in this process (the call it happened in returned NaN
// embeddings; later calls take the multi-launch path).  A host read of a pinned word: no synchronisation, callable any time --
// meaningful for a given forward once the stream it ran on has been synchronised.
extern "C" int vtc_cam_fused_gave_up(int device) {
  if (device < 0 || device >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_cam_mu);
  return g_cam_err_words[device] && *(volatile int *)g_cam_err_words[device] != 0 ? 1 : 0;
}
