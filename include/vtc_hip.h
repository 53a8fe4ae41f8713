/*
 * vtc_hip.h -- C ABI of libvtc_hip.so: the MI355X (gfx950) implementation of the VTC
 * retrieval forward/eval hot path.
 *
 * The reference (unitaryai/VTC) is pure Python: its native boundary for this path is
 * torch / openai-CLIP / faiss internals, it has no FFI of its own.  These entry points are
 * therefore what a binding for this path would call; each one names the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (so this header needs no HIP headers);
 *   - functions only enqueue work on `stream`; they never allocate, free or synchronise;
 *   - scratch memory is caller-provided: ask vtc_*_workspace_bytes(), pass `ws`/`ws_bytes`;
 *   - return 0 on success, non-zero on error (see vtc_last_error); nothing throws;
 *   - `dtype` selects the arithmetic of the GEMM/attention operands: VTC_F32 (exact fp32
 *     MFMA) or VTC_BF16 (bf16 operands, fp32 accumulate).  LayerNorm, softmax, residual
 *     stream, biases and all outputs are fp32 in both modes;
 *   - weight matrices are row-major [out_features, in_features] (PyTorch Linear layout) in
 *     the compute dtype; biases and LayerNorm parameters are fp32.
 *
 * Process-wide state: the entry points take no hidden arguments EXCEPT the environment variables below --
 * diagnostics and kernel-tuning knobs, each read ONCE into a function-local static at the first call that reaches it (C++11
 * thread-safe initialisation, never written afterwards), so they are fixed for the life of the process and invisible to the
 * per-call arguments.  Unset (the product configuration) every one of them selects the default.  None changes a result beyond
 * summation order; tests/test_abi.py holds this list to the getenv() calls of the sources.
 *   VTC_GEMM_TILE=1|2|4|5      force the GEMM kernel: 1 128 x 128, 2 256 x 256 free-running, 4 256 x 256 phased, 5 64 x 64 (default 0: by shape)   gemm.hip
 *   VTC_GEMM_DEEP=0|1          256 x 256 K loop: 1 (default) the deep LDS-DMA pipeline, 0 the round-3 loop; bit-identical
 *   VTC_GEMM_CG=n, VTC_GEMM_SUPER=n   tile walk: column-group width / row tiles per super-row (default 0 = by shape)
 *   VTC_GEMM_RESID_SMALL_K=k   residual GEMMs with K <= k on 128 x 128 tiles (two workgroups per CU; default 0: the heuristic alone)
 *   VTC_GEMM_STAGGER=g,t       start workgroup groups t ticks apart (a measured null, kept for A/B)
 *   VTC_GEMM_CU_BUDGET=n       the persistent 256 x 256 grid on at most n CUs (two streams' GEMMs side by side: tools/two_stream_halves.py)
 *   VTC_PATCH_IM2ROW=1         patch embedding through the im2row matrix instead of the in-place LDS-DMA gather
 *   VTC_SWEEP_MIN_TILE=0|1     distance GEMM of the block-minima sweep: 0 (default) 256 x 256 phased tiles, 1 128 x 128 tiles          sweep.hip
 *   VTC_SWEEP_EXACT_V1=1       the round-2 EXACT sweep (split-bf16 candidate lists) instead of the block-minima sweep
 *   VTC_SWEEP_PLANES=2|3|4     force the key-plane count of the recall-only sweep (default: by max k and size)
 *   VTC_SWEEP_DEBUG=1          stderr line per sweep with the fallback counters (synchronises)
 *   VTC_CAM_FUSED_MAX_ROWS=n   largest token count the one-launch CAM takes (default 512)                              cam.hip
 *   VTC_CAM_COOP=1             hipLaunchCooperativeKernel for the one-launch CAM (default: occupancy-checked ordinary launch)
 *   VTC_CAM_STAMPS=1           per-phase cycle stamps of the one-launch CAM on stderr (synchronises)
 * (VTC_CAM_TEST_GAVE_UP exists in the TEST build only: libvtc_hip_testhooks.so, -D VTC_TEST_HOOKS.)
 * The Python host layer's own switches (VTC_COMPUTE_DTYPE, VTC_OVERLAP, VTC_TEXT_RAGGED, VTC_TEXT_HALF_LAYERS, VTC_SWEEP_RANK,
 * VTC_CLIP_WEIGHTS, ...) are listed in INTEGRATION.md; they travel to the library as per-call arguments / per-model flags.
 */
#ifndef VTC_HIP_H
#define VTC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VTC_F32 = 0, VTC_BF16 = 1, VTC_U8 = 2 /* pixel_dtype only: raw 0..255 pixels */,
       VTC_F16 = 3 /* IEEE half operands, fp32 accumulate: vtc_gemm / vtc_layernorm / vtc_attention and the text tower */ };

/* residual activations of the CAM, model/model.py:65-77 (stateless ones) */
enum { VTC_ACT_NONE = 0, VTC_ACT_NORMALIZE = 1, VTC_ACT_SQUASH = 2, VTC_ACT_TANH = 3,
       VTC_ACT_SUB_MEAN = 4, VTC_ACT_BN = 5 };   /* eval-mode BatchNorm1d statistics, model/model.py:42-61 */

/* sweep precision: how q.g is formed from fp32 embeddings */
enum { VTC_SWEEP_F32 = 0,      /* fp32 MFMA, bitwise a k-ordered fmaf chain            */
       VTC_SWEEP_BF16X3 = 1,   /* hi/lo bf16 split, 3 products: |err| ~ 5e-7           */
       VTC_SWEEP_BF16 = 2,     /* plain bf16 operands: |err| ~ 1e-3, ranks may differ  */
       VTC_SWEEP_EXACT = 3 };  /* BF16X3 candidate lists (depth + 21, at least 32) re-ranked with
                                  fp64 distances sum_k (q_k - g_k)^2; a row whose candidate list is
                                  not provably a superset of its true top-`depth` is recomputed by
                                  fp64 brute force: ranks are those of exact fp64 arithmetic        */

/* One residual attention block.
 * upstream clip/model.py ResidualAttentionBlock; TimeSformer extras
 * model/timesformer_clip_alt.py:112-129 (NULL for ViT / text / CAM blocks). */
typedef struct {
  const float *ln1_g, *ln1_b;
  const void  *qkv_w;  const float *qkv_b;   /* attn.in_proj_{weight,bias}   [3W,W],[3W] */
  const void  *out_w;  const float *out_b;   /* attn.out_proj                [W,W],[W]   */
  const float *ln2_g, *ln2_b;
  const void  *fc_w;   const float *fc_b;    /* mlp.c_fc                     [4W,W],[4W] */
  const void  *proj_w; const float *proj_b;  /* mlp.c_proj                   [W,4W],[W]  */
  const float *lnt_g, *lnt_b;                /* ln_time                                  */
  const void  *tqkv_w; const float *tqkv_b;  /* timeattn.in_proj                         */
  const void  *tout_w; const float *tout_b;  /* timeattn.out_proj; NULL => tfc_* holds the
                                                pre-multiplied map temporal_fc o out_proj */
  const void  *tfc_w;  const float *tfc_b;   /* temporal_fc                  [W,W],[W]   */
  /* Folded LayerNorm (optional; all NULL => the LayerNorm kernels run).  16-bit modes only: the LayerNorm in front of a
   * projection is applied by that projection's epilogue,  LN(x) W^T + b = rstd_m (x W'^T - mean_m s_n) + c_n,  with
   * W' = (gamma . W) in the block's operand format, s[n] = sum_k W'[n][k] (of the rounded W'), c[n] = b[n] + sum_k beta[k] W[n][k].
   * Contract of the residual stream while the fold is on: between layer 0's cast and the final LayerNorm the stream is NOT the
   * fp32 `x` of the workspace (stale there) but a row-centred 16-bit pair (hi, lo) in the block's operand format, x - mean(x) ~=
   * hi + lo, which the residual GEMMs read and write (with per-row (mean, rstd)) and whose `hi` is the projections' A operand;
   * every buffer's rows are padded to a multiple of 256 and the pad rows hold garbage that only GEMM tiles touch. */
  const void  *qkv_wf;  const float *qkv_s,  *qkv_c;    /* ln_1    -> attn.in_proj     */
  const void  *fc_wf;   const float *fc_s,   *fc_c;     /* ln_2    -> mlp.c_fc         */
  const void  *tqkv_wf; const float *tqkv_s, *tqkv_c;   /* ln_time -> timeattn.in_proj */
} vtc_block_w;

/* Per-model path switches (vtc_vision_w.flags / vtc_text_w.flags).  They live in the weight struct -- two models in one
 * process may choose differently, nothing is process-wide -- and never change results beyond the stated tolerance
 * (FULL_LAST_LAYER: see below). */
enum { VTC_TOWER_NO_LN_FOLD = 1,       /* run the LayerNorm kernels even when the blocks carry folded weights (*_wf)              */
       /* 2, 4: retired in ABI 6 (the fused QKV + attention kernel of rounds 1-4 left the product: tools/probes/qkv_attn.hip) */
       VTC_TOWER_FULL_LAST_LAYER = 8,  /* By default the LAST block's queries, out_proj and MLP run only on the rows that reach the
                                          output (x[:, 0] behind ln_post, model/timesformer_clip_alt.py:281; the EOT row behind
                                          ln_final): on every other row of that block they are dead -- nothing reads them (its keys
                                          and values are computed for every row).  This flag computes them anyway (same embeddings
                                          within rounding; for measurements against the unpruned path)                          */
       VTC_TOWER_NO_SPLITK = 16        /* (ABI 7) At batch 1 - 2 (<= 1024 padded rows) the MLP's c_proj runs split over K: slices
                                          of K on 4x the workgroups, then one row pass that sums them, applies the residual update
                                          and writes the LayerNorm statistics.  This flag keeps the one-pass GEMM (same embeddings
                                          up to the summation order over K)                                                      */ };

/* Vision tower: upstream VisionTransformer (nframes == 0) or
 * model/timesformer_clip_alt.py:203-286 VisualTransformer (nframes > 0). */
typedef struct {
  int width, heads, layers, patch, grid, embed_dim, nframes;
  int variant;                    /* 0: timesformer_clip_alt.py (used by model.py); 1: timesformer_clip.py */
  int flags;                      /* VTC_TOWER_* (below): per-model choices of the kernel path, same results            */
  float pix_mean[3], pix_std[3];  /* pixel_dtype VTC_U8: x = (u8/255 - mean[c]) / std[c], i.e. ToTensor + Normalize of
                                     CLIP_TRANSFORM (dataset_loaders/dataset_loaders.py:40-49) fused into the patch gather */
  const void  *conv_w;            /* conv1.weight flattened [W, Kp], Kp = 3*patch*patch rounded up to a multiple of 64 (zero-padded
                                     columns: patch 14 -> 588 -> 640; patch 16 / 32 unchanged)   */
  const float *class_embedding;   /* [W]                                                  */
  const float *pos;               /* positional_embedding [1+grid*grid, W]                */
  const float *temporal;          /* temporal_embed [nframes, W] or NULL                  */
  const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
  const float *proj_t;            /* proj^T [embed_dim, W], fp32 in both modes            */
  const vtc_block_w *blocks;      /* HOST array of `layers` entries                       */
} vtc_vision_w;

/* Text tower: upstream CLIP.encode_text. */
typedef struct {
  int width, heads, layers, ctx, vocab, embed_dim;
  int half_layers;                /* dtype VTC_BF16 only: blocks [0, half_layers) hold IEEE-half (VTC_F16) weight matrices and
                                     run with half operands, the others bf16 (the bf16 rounding floor of
                                     this tower is above the 1e-3 budget; half has 3 more significant bits at the same rate) */
  int flags;                      /* VTC_TOWER_*                                          */
  const float *tok_emb;           /* token_embedding.weight [vocab, W] fp32               */
  const float *pos;               /* positional_embedding [ctx, W]                        */
  const float *ln_final_g, *ln_final_b;
  const float *proj_t;            /* text_projection^T [embed_dim, W], fp32 in both modes */
  const vtc_block_w *blocks;      /* HOST array                                           */
} vtc_text_w;

/* Context Adapter Module: model/model.py:141-205 + :396-400. */
typedef struct {
  int width, heads, layers;
  int init_from_avg;              /* model/model.py:156-161                               */
  int residual_activation;        /* VTC_ACT_*                                            */
  float squash_scale;             /* 1, 10, 1.2, 1.5, 1.8 for the squash* family          */
  const void  *final_linear;      /* [D,D] compute dtype (used when !init_from_avg)       */
  const float *mask_embedding;    /* [D]                                                  */
  const vtc_block_w *blocks;      /* HOST array                                           */
  const float *bn_mean, *bn_var;  /* mean_center_bn.running_mean / running_var [D]; SUB_MEAN, BN only (else NULL) */
  int flags;                      /* VTC_CAM_*                                            */
} vtc_cam_w;
/* vtc_cam_w.flags.  By default small batches (B (1 + nc) <= 512 tokens, fp32, init_from_avg) run the whole CAM -- token build,
 * both layers, finalisation -- as ONE cooperative launch (cam.hip) instead of ~30 generic ones; same arithmetic, fp32. */
enum { VTC_CAM_NO_FUSED = 1 };   /* always the multi-launch path */

const char *vtc_last_error(void);          /* thread-local message of the last failure   */
int vtc_abi_version(void);

/* ---- towers ------------------------------------------------------------------------ */

/* Replaces clip_model.encode_image (model/model.py:332,464) when w->nframes == 0 and
 * VisualTransformer.forward (model/timesformer_clip_alt.py:252-286, called at
 * model/model.py:497,613) when w->nframes > 0.
 * pixels: [n_items, F, 3, H, W] (F = 1 for images), fp32 (pixel_dtype VTC_F32), bf16, or raw uint8 (VTC_U8).
 * out:    [n_items, embed_dim] fp32 (not normalised). */
size_t vtc_vision_workspace_bytes(const vtc_vision_w *w, int n_items, int frames, int dtype);
int vtc_vision_forward(const vtc_vision_w *w, const void *pixels, int pixel_dtype, int n_items, int frames,
                       float *out, void *ws, size_t ws_bytes, int dtype, void *stream);

/* Replaces clip_model.encode_text (model/model.py:210,340,351,472,499,615).
 * ids: [n_seq, ctx] int64.  out: [n_seq, embed_dim] fp32. */
size_t vtc_text_workspace_bytes(const vtc_text_w *w, int n_seq, int dtype);
int vtc_text_forward(const vtc_text_w *w, const int64_t *ids, int n_seq, float *out, void *ws, size_t ws_bytes,
                     int dtype, void *stream);

/* The same tower over TWO id arrays -- sequences [0, n_a) from ids_a (titles), [n_a, n_a + n_b) from ids_b (comments; may be
 * NULL with n_b = 0) -- so that the wrappers need no concatenated copy (model/model.py:472 + :210 as one call), and, with
 * ragged != 0, on the ragged batch WITHOUT any host knowledge of the lengths: EOT positions, the prefix sums and the row count
 * are computed on the device and every kernel of the tower reads the row count from device memory (no host sync).
 * Workspace: vtc_text_workspace_bytes(w, n_a + n_b, dtype) (the dense upper bound).  Same outputs as vtc_text_forward. */
int vtc_text_forward2(const vtc_text_w *w, const int64_t *ids_a, int n_a, const int64_t *ids_b, int n_b, int ragged, float *out,
                      void *ws, size_t ws_bytes, int dtype, void *stream);

/* Same tower on a RAGGED batch: only tokens 0..EOT of each sequence are computed.  seq_offsets: int32
 * [n_seq+1] exclusive prefix sums of (argmax(ids[s]) + 1); total_rows = seq_offsets[n_seq] (host value).
 * Outputs are identical to vtc_text_forward: under the causal mask nothing after EOT reaches the EOT row. */
size_t vtc_text_ragged_workspace_bytes(const vtc_text_w *w, int n_seq, int total_rows, int dtype);
int vtc_text_forward_ragged(const vtc_text_w *w, const int64_t *ids, int n_seq, const int *seq_offsets, int total_rows,
                            float *out, void *ws, size_t ws_bytes, int dtype, void *stream);

/* Replaces PretrainedCLIPBase._load_comment_features' masking + _adapt_feature
 * (model/model.py:207-214, 141-205), eval semantics.
 * main: [B,D] fp32; comm_feats: [B*nc, D] fp32 = encode_text(comments.reshape(B*nc, ctx));
 * comments: [B,nc,ctx] int64 (only token 1 is read: the empty-string test :208).
 * adapted: [B,D] fp32 = normalize(normalize(main) + r). */
size_t vtc_cam_workspace_bytes(const vtc_cam_w *w, int B, int nc, int dtype);
int vtc_cam_forward(const vtc_cam_w *w, const float *main_feats, const float *comm_feats, const int64_t *comments,
                    int ctx, int B, int nc, float *adapted, void *ws, size_t ws_bytes, int dtype, void *stream);
/* Small batches run the whole module as ONE launch with software grid barriers (cam.hip).  A barrier that cannot complete --
 * a kernel this process cannot see holds CUs: another process on the card, a collective -- gives up after ~0.4 s: THAT call's
 * `adapted` rows are NaN, and every later call on the device takes the multi-launch path (same results; one line on stderr).
 * vtc_cam_fused_gave_up(device) = 1 once that has happened in this process (a host read of a pinned word, no synchronisation;
 * it speaks for a given forward once the forward's stream has been synchronised). */
int vtc_cam_fused_gave_up(int device);

/* ---- small fp32 ops of the wrappers (model/model.py:26-27, 338, 357-362, 369) -------- */
int vtc_normalize_rows(const float *x, float *out, int n, int d, void *stream);
/* Both embedding sets of a forward in one launch: outx = rows of x [nx, d] / their norms, outy likewise for y [ny, d] (the two
 * `normalize` calls that end every PretrainedCLIP*.forward, model/model.py:263-264, 366-367), and the non-finite watchdog with it:
 * flag[0] (int32, device; NULL: none) |= 1 when a row of x holds a NaN / inf, |= 2 for y (ABI 7; see vtc_nonfinite_flag2). */
int vtc_normalize_rows2(const float *x, float *outx, int nx, const float *y, float *outy, int ny, int d, int *flag, void *stream);
/* out[g] = mean over `group` consecutive rows (frames -> video, title+comments -> text) */
int vtc_mean_groups(const float *x, float *out, int n_groups, int group, int d, void *stream);
/* Token packing: the array-building half of `_tokenise` (dataset_loaders/dataset_loaders.py:224-248; the BPE encoder and the RAKE
 * summariser in front of it stay host text processing).  tokens: the batch's BPE ids back to back (int32, device), offsets [n_seq + 1]
 * their prefix sums (device).  ids [n_seq, ctx] int64 = [sot] + tokens of the sequence + [eot], zero padded; a sequence that reaches
 * ctx keeps its first ctx - 1 ids and ends in eot (:240-243).  The output is what vtc_text_forward* take. */
int vtc_pack_tokens(const int *tokens, const int *offsets, int n_seq, int ctx, int sot, int eot, int64_t *ids, void *stream);

/* out[g] = (a[g] + sum_k b[g * group + k]) / (1 + group): the "averaging" comment fusion, mean of a title embedding and its
 * comments' (model/model.py:357-362), without stacking them first */
int vtc_mean_head_groups(const float *a, const float *b, float *out, int n_groups, int group, int d, void *stream);
/* out[g] = mean of rows [offsets[g], offsets[g+1]): per-video mean over a ragged number of 8-frame
 * chunks, NOT re-normalised (evaluation/retrieval_evaluation.py:254-259).  offsets: int32 [n_groups+1] */
int vtc_segment_mean(const float *x, const int *offsets, float *out, int n_groups, int d, void *stream);
/* flag[0] |= 1 when x[0..n) holds a non-finite value.  Range guard of the text tower's IEEE-half blocks (vtc_text_w.half_layers):
 * an overflow of the half format (|v| > 65504) shows as inf / NaN in the tower's output.  `flag` is int32 in device memory or in
 * pinned (device-visible) host memory, which the host can then poll without synchronising. */
int vtc_nonfinite_flag(const float *x, size_t n, int *flag, void *stream);
/* flag[0] |= 1 when x[0..n) holds a non-finite value, |= 2 when y[0..m) does -- one launch over the two embedding sets a wrapper's
 * forward returns (ABI 7).  Every `PretrainedCLIP*.forward` ends with it (vtc_amd/host/model.py: the flag travels to pinned host
 * memory by an asynchronous copy and is read at the model's next forward / check_finite()); RecallAtK and the eval entry points
 * check their inputs with it and raise: a NaN embedding (an IEEE-half overflow in the text blocks, a one-launch CAM whose grid
 * barrier gave up, non-finite weights or pixels) never reaches a recall figure silently. */
int vtc_nonfinite_flag2(const float *x, size_t n, const float *y, size_t m, int *flag, void *stream);
/* sim[nv,nt] = exp(*logit_scale) * v @ t^T, fp32 exact */
int vtc_similarity(const float *v, const float *t, int nv, int nt, int d, const float *logit_scale, float *sim,
                   void *stream);

/* Replaces model/loss.py:18-22 clip_loss.  sim: [n,n] fp32; loss: 1 float (device). */
size_t vtc_clip_loss_workspace_bytes(int n);
int vtc_clip_loss(const float *sim, int n, float *loss, void *ws, size_t ws_bytes, void *stream);

/* ---- N x N sweep: replaces faiss.GpuIndexFlatL2.add/search in RecallAtK.compute
 * (model/metric.py:137-146) and the hit counting (:148-160) --------------------------- */

/* ids[nq,depth] (int64) / dists[nq,depth] (fp32, may be NULL) = the `depth` rows of `gallery`
 * nearest to each row of `queries` by squared L2 = |q|^2 + |g|^2 - 2 q.g (fp32), ascending,
 * ties by lowest gallery index.  The fp32 distance matrix is materialised in `ws` in
 * row blocks of `rows_per_block` queries (0 = as many as fit).
 * LIMITS (refused with a status + vtc_last_error, nothing is launched): d % 64 == 0 -- zero-pad the feature columns, squared L2
 * distances are unchanged (faiss.IndexFlatL2 takes any d; vtc_amd/host/metric.py pads for its callers) -- and
 * 1 <= depth <= min(64, n_gallery): one list entry per lane of a wavefront; the reference searches depth max(k)+1 = 11
 * (model/metric.py:145).  The same limits hold for vtc_l2_topk_bidir and the sharded entry points below. */
size_t vtc_l2_topk_workspace_bytes(int n_gallery, int n_queries, int d, int precision, int rows_per_block);
int vtc_l2_topk(const float *gallery, const float *queries, int n_gallery, int n_queries, int d, int depth,
                int precision, int rows_per_block, int64_t *ids, float *dists, void *ws, size_t ws_bytes,
                void *stream);

/* Both retrieval directions from ONE distance matrix D[i][j] = |b_i - a_j|^2 (the reference runs two searches,
 * evaluation/eval.py:117-127 -> model/metric.py:137-146): ids_b2a [n_b, depth] = vtc_l2_topk(gallery = a, queries = b),
 * ids_a2b [n_a, depth] = vtc_l2_topk(gallery = b, queries = a); the second direction is read off the columns of the
 * blocks the first direction's GEMM has written (no second GEMM).  dists_* may be NULL.  depth <= min(64, n_a, n_b). */
size_t vtc_l2_topk_bidir_workspace_bytes(int n_a, int n_b, int d, int precision, int rows_per_block);
int vtc_l2_topk_bidir(const float *a, const float *b, int n_a, int n_b, int d, int depth, int precision, int rows_per_block,
                      int64_t *ids_b2a, float *dists_b2a, int64_t *ids_a2b, float *dists_a2b, void *ws, size_t ws_bytes,
                      void *stream);
/* Sharded sweep (one process per GPU; the reference is single-GPU, evaluation/eval.py:117-127 -> model/metric.py:137-146):
 * rank r holds rows [lo_r, hi_r) of both embedding sets and the all-gathered sets, and runs ONE distance GEMM
 * D_r[i][j] = |b_(lo_r + i) - a_j|^2, i < n_local, j < n_total, for BOTH directions:
 *   vtc_l2_sweep_shard_rows   ids [n_local, depth] = vtc_l2_topk(gallery = a_all, queries = b_local) -- complete, the rank
 *                             holds whole rows -- and col_planes [4, nblk_pad, n_total] (uint32): per column j and per block of
 *                             vtc_l2_sweep_row_block() local rows, the three smallest distance keys and the fourth as a bound
 *                             (blocks past the rank's rows: +inf).  nblk_pad = ceil(largest shard / row block), equal on all ranks.
 *   (host) all-to-all: rank r sends col_planes[:, :, lo_s:hi_s] to rank s
 *   vtc_l2_sweep_shard_cols   planes [n_src, 4, nblk_pad, n_local] = what the n_src ranks sent for this rank's columns, in
 *                             rank order; src_base[n_src] (device int32) = lo_r of each source.  ids [n_local, depth] =
 *                             vtc_l2_topk(gallery = b_all, queries = a_local), bit-identical to the single-GPU search
 *                             (certified candidates re-ranked in fp64, uncertified columns by fp64 brute force).
 * VTC_SWEEP_EXACT only; d % 64 == 0; vtc_l2_sweep_shard_supported tells whether the shape is covered (else: two vtc_l2_topk). */
int vtc_l2_sweep_row_block(void);
int vtc_l2_sweep_shard_supported(int n_total, int n_local, int depth);
size_t vtc_l2_sweep_shard_workspace_bytes(int n_total, int n_local, int d);
int vtc_l2_sweep_shard_rows(const float *a_all, const float *b_local, int n_total, int n_local, int d, int depth, int64_t *ids,
                            float *dists, unsigned *col_planes, int nblk_pad, void *ws, size_t ws_bytes, void *stream);
int vtc_l2_sweep_shard_cols(const float *b_all, const float *a_local, int n_total, int n_local, int d, int depth,
                            const unsigned *planes, int n_src, int nblk_pad, const int *src_base, int64_t *ids, float *dists,
                            void *ws, size_t ws_bytes, void *stream);
/* hits[j] += #{ i : (target_offset + i) in ids[i, :k_vals[j]] }   (hits: int64 device) */
int vtc_recall_hits(const int64_t *ids, int n_queries, int depth, int64_t target_offset, const int *k_vals_host,
                    int nk, long long *hits, void *stream);
/* R@K of BOTH directions of n PAIRED rows (a_i <-> b_i: RecallAtK.result(), model/metric.py:166-187; evaluation/eval.py:117-127) without the
 * sorted neighbour lists (round 5).  The reference asks one thing of its search -- is the query's own index among the first k -- which is
 * the RANK of one gallery row: #{ j : (|q - g_j|^2, j) < (|q - g_t|^2, t) } < k.  One prologue, ONE distance GEMM (the block-minima
 * planes of vtc_l2_topk_bidir), ONE rank launch: entries whose key is more than the row's error bound away from the target's exact
 * distance are counted or dropped unseen, the few in reach are settled in fp64.  The counters are those of vtc_l2_topk_bidir(depth =
 * max k + 1, VTC_SWEEP_EXACT) followed by vtc_recall_hits_pair, exactly.  hits_*: nk device int64 each, ADDED to:
 *   hits_b_from_a[j] += #{ i : a_i is among the k_j nearest a's of b_i }   = RecallAtK.compute(a, b) x n
 *   hits_a_from_b[j] += #{ i : b_i is among the k_j nearest b's of a_i }   = RecallAtK.compute(b, a) x n
 * n >= 1024, d % 64 == 0 (vtc_l2_recall_bidir_supported; else the two-call form); nk <= 4; k_vals on the host.
 * Non-finite embeddings (ABI 7): a pair whose target distance |a_i - b_i|^2 is NaN / inf -- any NaN or inf in a_i or b_i -- is a MISS at
 * every k (an exact search never returns such a target), and the call ORs VTC_RECALL_NONFINITE (bit 40) into hits_b_from_a[0] and
 * hits_a_from_b[0]: the caller learns of it from the counters it reads anyway (counter = value & (VTC_RECALL_NONFINITE - 1); the host
 * layer raises).  The same holds for vtc_l2_recall_shard_rows / _cols (bit 40 survives the sum over <= 2^22 ranks). */
#define VTC_RECALL_NONFINITE (1ll << 40)
int vtc_l2_recall_bidir_supported(int n, int d);
size_t vtc_l2_recall_bidir_workspace_bytes(int n, int d);
int vtc_l2_recall_bidir(const float *a, const float *b, int n, int d, const int *k_vals_host, int nk, long long *hits_b_from_a,
                        long long *hits_a_from_b, void *ws, size_t ws_bytes, void *stream);
/* The sharded sweep (above) with the recall-only finish: rank r holds rows [row_base, row_base + n_local) of both sets and the gathered
 * sets, runs ONE [n_local, n_total] distance GEMM and adds its PARTIAL counters (the host all-reduces them: model/metric.py:148-160 counted
 * over this rank's queries):
 *   vtc_l2_recall_shard_rows   hits_b_from_a[j] += #{ i local : a_(row_base + i) among the k_j nearest a's of b_(row_base + i) };
 *                              col_planes [P, nblk_pad, n_total] in vtc_l2_sweep_shard_rows' layout, P = vtc_l2_recall_planes(k_vals, nk, n_total):
 *                              for max k <= 16 two planes (per column and block the smallest key + the second as a bound; n_total >= 16 384)
 *                              or three (two keys + bound), else four
 *   (host) all-to-all of the column planes, as above
 *   vtc_l2_recall_shard_cols   planes [n_src, P, nblk_pad, n_local]; src_bounds [n_src + 1] (device int32): source s ran rows
 *                              [src_bounds[s], src_bounds[s + 1]);  hits_a_from_b[j] += #{ i local : b_(row_base + i) among the k_j nearest
 *                              b's of a_(row_base + i) }
 * Summed over the ranks: the counters of vtc_l2_recall_bidir on the gathered sets, exactly.  n_total >= 1024, d % 64 == 0, nk <= 4;
 * workspace: vtc_l2_sweep_shard_workspace_bytes. */
int vtc_l2_recall_shard_supported(int n_total, int n_local, int d);
int vtc_l2_recall_planes(const int *k_vals_host, int nk, int n_total);
int vtc_l2_recall_shard_rows(const float *a_all, const float *b_local, int n_total, int n_local, int row_base, int d,
                             const int *k_vals_host, int nk, long long *hits_b_from_a, unsigned *col_planes, int nblk_pad, void *ws,
                             size_t ws_bytes, void *stream);
int vtc_l2_recall_shard_cols(const float *b_all, const float *a_local, int n_total, int n_local, int row_base, int d,
                             const int *k_vals_host, int nk, const unsigned *planes, int n_src, int nblk_pad, const int *src_bounds,
                             long long *hits_a_from_b, void *ws, size_t ws_bytes, void *stream);
/* the same for the two directions of one evaluation (RecallAtK.result(), model/metric.py:166-187: compute(a, b) and compute(b, a),
 * same number of queries and the same targets) in ONE launch */
int vtc_recall_hits_pair(const int64_t *ids_a, const int64_t *ids_b, int n_queries, int depth, int64_t target_offset,
                         const int *k_vals_host, int nk, long long *hits_a, long long *hits_b, void *stream);

/* ---- primitives (exported for parity tests and reuse) -------------------------------- */
enum { VTC_EPI_STORE = 0,   /* out = acc + bias                      (out dtype = out_dtype) */
       VTC_EPI_GELU = 1,    /* out = QuickGELU(acc + bias)                                     */
       VTC_EPI_RESID = 2 }; /* out(fp32) += acc + bias; rows with m % skip_mod == 0 untouched  */
/* out[M,N] = epi(A[M,K] @ W[N,K]^T + bias).  A, W in `dtype`; K % 64 == 0 (bf16) / % 32 (fp32). */
int vtc_gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype,
             int epilogue, int out_dtype, int skip_mod, void *stream);
/* Residual GEMM + the LayerNorm that follows it, one launch (16-bit operand formats; N a multiple of 256, <= 1024; problems
 * large enough for the 256 x 256 kernel -- ask vtc_gemm_resid_layernorm_supported):  out(fp32) += A W^T + bias with rows
 * m % skip_mod == 0 untouched, then ln_out[M,N] (operand format) = LayerNorm(out) * ln_g + ln_b.  Bit-identical to
 * vtc_gemm(VTC_EPI_RESID) + vtc_layernorm.  Replaces e.g. `x = x + attn(...)` followed by `ln_2(x)`
 * (model/timesformer_clip_alt.py:173-174).  ln_out may alias A (a row block's A rows are dead when its LayerNorm runs). */
size_t vtc_gemm_resid_layernorm_workspace_bytes(int M);
int vtc_gemm_resid_layernorm_supported(int M, int N, int K, int dtype);
int vtc_gemm_resid_layernorm(const void *A, const void *W, const float *bias, float *out, int M, int N, int K, int dtype,
                             int skip_mod, const float *ln_g, const float *ln_b, void *ln_out, void *ws, size_t ws_bytes,
                             void *stream);
/* y[r,:] = LN(x[row(r),:]) * g + b, row(r) = row_index ? row_index[r] : r * row_mul; out dtype selectable */
int vtc_layernorm(const float *x, const float *g, const float *b, void *y, int rows, int width, int out_dtype,
                  const int *row_index, int row_mul, void *stream);
/* softmax(q k^T / sqrt(64) [+causal]) v per (sequence, head) on a packed qkv buffer [rows, 3W];
 * token p of sequence s lives at row  (s / s2) * a1 + (s % s2) * a2 + a0 + (p ? 1 + (s % s2) * a3 + (p - 1) * pstride : 0).
 * out: [rows, W] (same row map).  If cls_out != NULL the p == 0 output goes to cls_out[s, W] (fp32) instead. */
int vtc_attention(const void *qkv, void *out, float *cls_out, int n_seq, int L, int heads, int causal, int s2, int a0,
                  int a1, int a2, int a3, int pstride, int dtype, void *stream);

/* ONE query per sequence (the last block of a tower, where only the row that reaches the output asks): sequence o attends
 * with the projected query q[o / s2] (q: [n_q, W], operand format, compact) over its keys and values in the packed qkv buffer (key rows:
 * the map of vtc_attention; the Q third of qkv is not read).  eot != NULL instead: the text tower's rows base .. eot[o], base =
 * offs ? offs[o] : o * ctx, query q[o].  out [n_out, W] fp32.  fp32 arithmetic throughout. */
int vtc_single_query_attention(const void *qkv, const void *q, float *out, int n_out, int L, int heads, int s2, int a0, int a1, int a2,
                               int a3, int pstride, const int *eot, const int *offs, int ctx, int dtype, void *stream);

/* ---- adapter-only training step: backward + optimizer primitives, fp32 -------------------
 * Replace, for PretrainedCLIP_finaltf with frozen towers (configs/pretrained_clip_comments_attn_frozen.jsonc), what
 * torch.autograd does behind `loss.backward()` (trainer/trainer.py) for clip_loss (model/loss.py:18-22), normalize
 * (model/model.py:26-27) and the CAM transformer (clip.model.Transformer, model/model.py:396-398), and
 * torch.optim.Adam(amsgrad=True).step().  Orchestrated by vtc_amd/host/adapter_train.py; dgrad / wgrad matrix
 * products are vtc_gemm calls on transposed operands. */
int vtc_transpose_f32(const float *x, float *y, int rows, int cols, void *stream);            /* y[c][r] = x[r][c] */
int vtc_colsum_f32(const float *x, float *out, int rows, int cols, void *stream);             /* bias gradients     */
/* dx (+)= LN'(x; gamma)(dy), dgamma = sum_r dy xhat, dbeta = sum_r dy (eps 1e-5, biased variance) */
int vtc_layernorm_bwd(const float *x, const float *gamma, const float *dy, float *dx, float *dgamma, float *dbeta, int rows,
                      int width, int accumulate_dx, void *stream);
/* unmasked attention backward, L <= 16, head_dim 64, sequences contiguous: qkv [n_seq*L, 3W], dout [n_seq*L, W] */
int vtc_attention_small_bwd(const float *qkv, const float *dout, float *dqkv, int n_seq, int L, int heads, void *stream);
/* QuickGELU: dy == NULL -> out = x sigmoid(1.702 x); else out = dy * d/dx */
int vtc_quickgelu(const float *x, const float *dy, float *out, size_t n, void *stream);
int vtc_normalize_rows_bwd(const float *x, const float *dy, float *dx, int n, int d, void *stream);
/* dsim = d clip_loss / d sim; ws >= 4 n floats */
int vtc_clip_loss_bwd(const float *sim, int n, float *dsim, void *ws, size_t ws_bytes, void *stream);
/* torch.optim.Adam single-tensor step (weight_decay 0); step counts from 1; vmax only when amsgrad */
int vtc_adam_step(float *p, const float *g, float *m, float *v, float *vmax, size_t n, float lr, float beta1, float beta2,
                  float eps, int step, int amsgrad, void *stream);
int vtc_axpby(float *out, const float *x, const float *y, float a, float b, size_t n, void *stream);   /* out = a x + b y  */
int vtc_scale_rows(float *x, const float *s, int rows, int d, int group, void *stream);               /* x[r] *= s[r/group] */

/* ---- optional per-launch timing (HIP events on the launch stream; bench/diagnostics) ----
 * Between vtc_prof_begin() and vtc_prof_end() every kernel launch of the library is bracketed by
 * two events.  vtc_prof_end synchronises `stream` and returns, per class, the summed kernel time,
 * the launch count and the summed work (FLOPs = 2MNK for GEMM, 4*L*L*64 per (sequence, head) for
 * attention; bytes moved for the others).  Process-global, single-threaded use only. */
enum { VTC_PROF_GEMM_BF16 = 0, VTC_PROF_GEMM_F32 = 1, VTC_PROF_ATTN = 2, VTC_PROF_NORM = 3, VTC_PROF_EMBED = 4,
       VTC_PROF_TOPK = 5, VTC_PROF_NCLASS = 6 };
/* Launches also carry the region of the tower they belong to (per thread): the attention branches (LayerNorm + QKV
 * projection + attention core + output projection [+ temporal_fc] + cls bookkeeping -- what BASELINE.md calls
 * "TimeSformer attention" for the video tower), the MLP branches, everything else. */
enum { VTC_PROF_REGION_OTHER = 0, VTC_PROF_REGION_ATTN = 1, VTC_PROF_REGION_MLP = 2, VTC_PROF_NREGION = 3 };
int vtc_prof_begin(void);
int vtc_prof_end(void *stream, double *ms, long long *launches, double *work);
/* as vtc_prof_end, split by region: arrays of VTC_PROF_NCLASS * VTC_PROF_NREGION, index cls * VTC_PROF_NREGION + region */
int vtc_prof_end_regions(void *stream, double *ms, long long *launches, double *work);
/* as vtc_prof_end, one record per launch (at most `max`; returns the count through *n): class, region, time, work and, for
 * GEMM launches, tag[3 i ..] = (epilogue mode, N, K) -- bench.py groups them to report the dominant single instantiation */
int vtc_prof_end_records(void *stream, int max, int *n, int *cls, int *region, double *ms, double *work, int *tag);
/* kernel launches made by the library since it was loaded (every launch site counts; diagnostics / bench) */
long long vtc_debug_launch_count(void);

#ifdef __cplusplus
}
#endif
#endif /* VTC_HIP_H */
