// towers.hip -- host-side orchestration of the three towers and the CAM: each entry point
// enqueues the kernel sequence of one forward pass on the caller's stream, using only the
// caller's workspace.  No allocation, no synchronisation, no global state.
//
//   vtc_vision_forward  upstream VisionTransformer.forward (nframes == 0) /
//                       model/timesformer_clip_alt.py:252-286 + block :135-175 (nframes > 0)
//   vtc_text_forward    upstream CLIP.encode_text
//   vtc_cam_forward     model/model.py:141-205 (+ the masking of :207-214)
//
// Activation layout in HBM (per call):
//   x    fp32 [rows, W]    residual stream, rows in the REFERENCE's token order
//                          (video: item-major, row item*T = cls, row item*T + 1 + n*F + t = patch n
//                          of frame t, timesformer_clip_alt.py:271-275), updated in place by the
//                          GEMM residual epilogues;
//   h    T    [rows, W]    LayerNorm output, then reused for the attention output;
//   big  T    [rows, 4W]   packed qkv ([rows,3W]), then the MLP hidden ([rows,4W]), and before the
//                          first block the im2row patch matrix;
// where T is the compute dtype (bf16 or fp32).  The reference's ~10 rearrange/cat/repeat copies
// per block (timesformer_clip_alt.py:143-173) do not exist here: the attention kernel addresses
// tokens through an affine row map and the cls bookkeeping is two tiny kernels.
#include <stdarg.h>
#include <string.h>

#include "common.h"
namespace vtcgemm { int num_cus(); }

static thread_local char g_err[512] = "";
void vtc_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *vtc_last_error(void) { return g_err; }
extern "C" int vtc_abi_version(void) { return 7; }
int launch_im2row(const void *px, int pixel_dtype, void *out, int dtype, int n_frames, int grid, int patch, int res, const float *mean, const float *stdv, hipStream_t stream);
int patch_k_padded(int patch);
int launch_pixels_u8_to_operand(const void *px, void *out, int dtype, int n_frames, int res, const float *mean, const float *stdv, hipStream_t stream);
int launch_cls_rows(float *x, const float *cls, const float *pos0, int n_items, int T, int W, hipStream_t stream);
int launch_cls_mean(const float *cls_tmp, void *out, int dtype, int n_items, int F, int T, int W, hipStream_t stream);
int launch_text_prep(const TextIds &ids, int n_seq, int ctx, int *lens, int *offsets, int *m_dev, hipStream_t stream);
int launch_text_embed(const TextIds &ids, const float *tok, const float *pos, float *x, int *eot_row, int n_seq, int ctx, int W, int vocab, hipStream_t stream);
int launch_text_embed_ragged(const TextIds &ids, const float *tok, const float *pos, const int *seq_offsets, float *x, int *eot_row, int n_seq, int ctx, int W, int vocab, hipStream_t stream);
int launch_attention_ragged(const void *qkv, void *out, int n_seq, int max_L, int heads, int causal, const int *seq_offsets, double flops, const int *rows_dev, int dtype, hipStream_t stream);
int launch_attention_generic_small(const void *qkv, void *out, int n_seq, int L, int heads, int hd, int dtype, hipStream_t stream);
int launch_cam_tokens(const float *main_f, const float *comm, const int64_t *comments, const float *mask_emb, float *X, int B, int nc, int ctx, int D, hipStream_t stream);
int launch_cls_global_attention(const void *qkv, void *out, int n_items, int Ttok, int heads, int dtype, hipStream_t stream);
int launch_cam_finalize(const float *Y, const float *lin, const float *main_f, float *out, int B, int Lc, int D, int init_from_avg, int act, float scale, const float *bn_mean, const float *bn_var, hipStream_t stream);

bool cam_fused_supported(const vtc_cam_w *w, int B, int nc, int dtype);
size_t cam_fused_bar_bytes();
int launch_cam_fused(const vtc_cam_w *w, const float *main_feats, const float *comm_feats, const int64_t *comments, int ctx, int B, int nc,
                     float *adapted, float *x, float *big, float *att, int *bar, hipStream_t stream);
namespace {

// (Rounds 1-4 carried a fused QKV projection + attention-core kernel behind per-model flags; it measured slower than GEMM + core in
// every configuration -- in round 5 also with the folded LayerNorm kept, on the time branch alone: 2.09 ms against 1.20 + 0.49 per
// layer at 1 024 videos, profiles/r05_experiments.txt 1 -- and left the product: tools/probes/qkv_attn.hip.)

struct Bump {
  char *base;
  size_t off = 0;
  explicit Bump(void *b) : base((char *)b) {}
  void *take(size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return base ? base + o : nullptr;
  }
};

#define RUN(call)        \
  do {                   \
    int rc_ = (call);    \
    if (rc_) return rc_; \
  } while (0)

inline int esz(int dtype) { return dtype == VTC_F32 ? 4 : 2; }

int gemm(const void *A, const void *W, const float *bias, void *out, int M, int N, int K, int dtype, int mode, int out_dtype,
         int skip_mod, hipStream_t s) {
  GemmEpi e;
  e.mode = mode; e.out_dtype = out_dtype; e.skip_mod = skip_mod;
  return launch_gemm(A, W, bias, out, M, N, K, dtype, e, s);
}

// ---- folded LayerNorm (include/vtc_hip.h vtc_block_w; GemmEpi::fold_*) ---------------------------------------------------
// The three LayerNorms of a block each sit between a residual GEMM (which has the updated row in registers) and a projection.
// Folded: the residual GEMM also writes the row in the operand format (`xb`) and per-(row, 64 columns) statistics; the
// projection reads xb against gamma-scaled weights and applies mean / rstd in its epilogue.  The LayerNorm kernels' 1.2 GB
// read + 0.6 GB write per launch (config 3, 1 024 videos) become a 0.6 GB write in the residual epilogue.  Needs every tile
// interior: rows padded to 256 (the pad rows hold garbage that no kernel outside the GEMMs reads), W a multiple of 256.
// flags & VTC_TOWER_NO_LN_FOLD: the LayerNorm kernels.
struct Fold {
  bool on = false;
  void *xb = nullptr;       // [rows_pad, W] operand format: the residual stream as the projections read it (hi of the pair)
  void *xl = nullptr;       // [rows_pad, W] operand format: lo of the pair -- between layer 0's cast and the final merge the
                            // stream lives in (xb, xl) and the fp32 x is stale (GemmEpi::y16lo)
  float *part = nullptr;    // [W / 64][rows_pad] (sum, squared deviations)
  float *stat = nullptr;    // [rows_pad] (mean, rstd)
  int rows_pad = 0;
  int fmt = -1;             // operand format (xb, xl) / stat currently hold (-1: the stream is the fp32 x)
  float *splitk = nullptr;  // [SPLITK_MAX][rows_pad, W] fp32 partial products of a split-K c_proj (rows_pad <= SPLITK_MAX_ROWS; else NULL)
};
// Split-K for the MLP's c_proj at batch 1 - 2: its 64 x 64 tiles are few (video, B = 1: 8 x 12 = 96) and each walks
// K = 3072 alone -- 24 us of a 109 us layer.  Cut into K slices it is one round of ~4x the workgroups, and the pass that sums the slices
// also applies the residual update and writes the row statistics whole: c_proj + fold_stats (24 + 5 us) become split GEMM + row pass.
constexpr int SPLITK_MAX = 4, SPLITK_MAX_ROWS = 1024;
// slices for a [rows_pad, W] x K residual GEMM: the largest S <= SPLITK_MAX with K % (64 S) == 0, at least four K-steps per slice, and
// every (tile, slice) resident at once (two 64 x 64 workgroups per CU); 1 = no split
inline int splitk_slices(int rows_pad, int W, int K) {
  if (rows_pad > SPLITK_MAX_ROWS || K < 2048 || W % 256 != 0 || W > 1024) return 1;
  const long tiles = (long)(rows_pad / 64) * (W / 64);
  for (int s = SPLITK_MAX; s >= 2; --s)
    if (K % (64 * s) == 0 && K / s >= 256 && tiles * s <= 2L * vtcgemm::num_cus()) return s;
  return 1;
}
// the fp32 rows a final LayerNorm reads (rows i * row_mul, or row_index[i]) out of the pair
int fold_merge_rows(Fold &f, float *x, int n, int W, const int *row_index, int row_mul, hipStream_t s) {
  if (!f.on || f.fmt < 0) return 0;
  return launch_split_merge_rows(f.xb, f.xl, x, n, W, row_index, row_mul, f.fmt, s);
}
inline int pad256(int rows) { return (rows + 255) / 256 * 256; }
bool fold_usable(const vtc_block_w *blocks, int layers, int W, int dtype, bool timesformer, int flags) {
  if ((flags & VTC_TOWER_NO_LN_FOLD) || dtype == VTC_F32 || W % 256 != 0) return false;
  for (int l = 0; l < layers; ++l)
    if (!blocks[l].qkv_wf || !blocks[l].fc_wf || (timesformer && !blocks[l].tqkv_wf)) return false;
  return true;
}

// A row count: known to the host (dev == NULL), or living in device memory -- the ragged text tower of vtc_text_forward2, whose
// lengths the host never sees: n is then the dense upper bound (it sizes grids and the workspace) and every kernel reads
// dev[0] = the rows, dev[1] = the rows padded to 256 (what the folded GEMMs run over).
struct Rows {
  int n;
  const int *dev = nullptr;
  Rows(int n_) : n(n_) {}
  Rows(int n_, const int *dev_) : n(n_), dev(dev_) {}
  const int *dev_pad() const { return dev ? dev + 1 : nullptr; }
};

// the stream as the pair (xb, xl) + row statistics in operand format `dtype` (no residual GEMM of this format in front: layer 0, or
// a format boundary -- back through fp32)
int fold_enter(Fold &f, const float *x, const Rows &rows, int W, int dtype, hipStream_t s) {
  if (f.fmt == dtype) return 0;
  if (f.fmt >= 0) RUN(launch_split_merge_rows(f.xb, f.xl, const_cast<float *>(x), rows.n, W, nullptr, 1, f.fmt, s, rows.dev));
  RUN(launch_cast_rowstats(x, f.xb, f.xl, f.stat, rows.n, W, dtype, s, rows.dev));
  f.fmt = dtype;
  return 0;
}

// out = epi(LN(x; g, bt) w^T + bias)
// (col0, ncols: only the output columns [col0, col0 + ncols) -- rows col0.. of the weight -- are computed, at their places in the
// N-wide output: the K and V thirds of a QKV projection whose queries come from elsewhere; ncols == 0: all N)
int ln_proj(Fold &f, const float *x, const float *g, const float *bt, const void *w, const float *bias, const void *wf, const float *fs,
            const float *fc, void *h, void *out, const Rows &rows, int N, int W, int dtype, int mode, hipStream_t s, int col0 = 0,
            int ncols = 0) {
  const int ldo = N;
  if (ncols > 0) {
    const size_t wo = (size_t)col0 * W * esz(dtype);
    w = (const char *)w + wo;
    if (wf) wf = (const char *)wf + wo;
    if (bias) bias += col0;
    if (fs) fs += col0;
    if (fc) fc += col0;
    out = (char *)out + (size_t)col0 * esz(dtype);
    N = ncols;
  }
  if (f.on) {
    RUN(fold_enter(f, x, rows, W, dtype, s));
    GemmEpi e;
    e.mode = mode; e.out_dtype = dtype; e.fold_stat = f.stat; e.fold_s = fs; e.m_dev = rows.dev_pad(); e.ldo = ldo;
    return launch_gemm(f.xb, wf, fc, out, f.rows_pad, N, W, dtype, e, s);
  }
  RUN(launch_layernorm(x, g, bt, h, rows.n, W, dtype, nullptr, 1, false, s, rows.dev));
  GemmEpi e;
  e.mode = mode; e.out_dtype = dtype; e.m_dev = rows.dev; e.ldo = ldo;
  return launch_gemm(h, w, bias, out, rows.n, N, W, dtype, e, s);
}

// x += A w^T + bias  (rows with m % skip_mod == 0 untouched)
// center: this update also subtracts the rows' previous means (the stream stays centred; once per layer is enough -- a single
// update moves a row's mean by a fraction of its spread)
int resid_proj(Fold &f, const void *A, const void *w, const float *bias, float *x, const Rows &rows, int W, int K, int dtype, int skip_mod,
               hipStream_t s, bool center = false, int flags = 0) {
  if (f.on) {
    GemmEpi e;
    e.mode = VTC_EPI_RESID; e.out_dtype = VTC_F32; e.skip_mod = skip_mod; e.y16 = f.xb; e.y16lo = f.xl; e.fold_part = f.part;
    e.m_dev = rows.dev_pad();
    if (center) e.fold_stat = f.stat;       // the rows' means before this update: the stream is stored centred (gemm.hip, SPLIT)
    if (f.fmt != dtype) {       // the stream is not a pair of this format yet (cannot happen behind ln_proj; kept for safety)
      if (f.fmt >= 0) RUN(launch_split_merge_rows(f.xb, f.xl, x, rows.n, W, nullptr, 1, f.fmt, s, rows.dev));
      RUN(launch_cast_rowstats(x, f.xb, f.xl, f.stat, rows.n, W, dtype, s, rows.dev));
      f.fmt = dtype;
    }
    const int nsl = (center && skip_mod == 0 && f.splitk && !(flags & VTC_TOWER_NO_SPLITK)) ? splitk_slices(f.rows_pad, W, K) : 1;
    if (nsl > 1) {
      GemmEpi ek;
      ek.mode = VTC_EPI_STORE; ek.out_dtype = VTC_F32; ek.m_dev = rows.dev_pad(); ek.ksplit = nsl; ek.split_stride = f.rows_pad * W;
      RUN(launch_gemm(A, w, nullptr, f.splitk, f.rows_pad, W, K, dtype, ek, s));
      return launch_splitk_resid_rows(f.splitk, nsl, f.rows_pad * W, bias, f.xb, f.xl, f.stat, f.rows_pad, W, dtype, s, rows.dev_pad());
    }
    RUN(launch_gemm(A, w, bias, x, f.rows_pad, W, K, dtype, e, s));
    return launch_fold_stats(f.part, W / 64, f.rows_pad, f.stat, s, rows.dev_pad());
  }
  GemmEpi e;
  e.mode = VTC_EPI_RESID; e.out_dtype = VTC_F32; e.skip_mod = skip_mod; e.m_dev = rows.dev;
  return launch_gemm(A, w, bias, x, rows.n, W, K, dtype, e, s);
}

// x += MLP(ln_2 x)   (timesformer_clip_alt.py:174 / upstream block)
int mlp_part(Fold &f, const vtc_block_w &b, float *x, void *h, void *big, const Rows &rows, int W, int dtype, hipStream_t s, int flags = 0) {
  ProfRegion region(VTC_PROF_REGION_MLP);
  RUN(ln_proj(f, x, b.ln2_g, b.ln2_b, b.fc_w, b.fc_b, b.fc_wf, b.fc_s, b.fc_c, h, big, rows, 4 * W, W, dtype, VTC_EPI_GELU, s));
  RUN(resid_proj(f, big, b.proj_w, b.proj_b, x, rows, W, 4 * W, dtype, 0, s, true, flags));
  return 0;
}

// x += MHA(ln_1 x) over n_seq contiguous sequences of L tokens
// (tail_src: stop behind the attention core and say where its output lies -- the last block, whose out_proj runs on the output
// rows only: last_block_tail)
int attn_part_contig(Fold &f, const vtc_block_w &b, float *x, void *h, void *big, int n_seq, int L, int W, int heads, int causal,
                     int dtype, int flags, hipStream_t s, const void **tail_src = nullptr) {
  const int rows = n_seq * L;
  ProfRegion region(VTC_PROF_REGION_ATTN);
  RUN(ln_proj(f, x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, h, big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s));
  if (W != heads * 64) {      // the CAM with head_dim != 64 (ViT-L/14's 768-d features at the reference's default n_heads = 8)
    VTC_CHECK(!causal && W % heads == 0, "attention: head_dim %d/%d", W, heads);
    RUN(launch_attention_generic_small(big, h, n_seq, L, heads, W / heads, dtype, s));
  } else {
    RUN(launch_attention(big, h, nullptr, n_seq, L, heads, causal, 1, 0, L, 0, 0, 1, dtype, s));
  }
  if (tail_src) { *tail_src = h; return 0; }
  RUN(resid_proj(f, h, b.out_w, b.out_b, x, rows, W, W, dtype, 0, s));
  return 0;
}

// ---- the last block: only the rows that reach the output --------------------------------------------------------------------
// Behind the last block a tower reads ONE row per item (x[:, 0] into ln_post, model/timesformer_clip_alt.py:281 and
// model/timesformer_clip.py:433; the EOT row into ln_final, upstream CLIP encode_text).  That block's attention still needs K and V
// of every row, but its out_proj, its residual update and its whole MLP are per-row maps: on any other row they produce values
// nothing ever reads.  So after the last attention core the n output rows are gathered into a compact fp32 stream and
// out_proj / ln_2 / c_fc / QuickGELU / c_proj run on those n rows only (LayerNorm kernels, the reference's own order of
// operations).  Exact, not an approximation: every value that can influence an embedding is computed as before.
// flags & VTC_TOWER_FULL_LAST_LAYER computes the dead rows anyway.
struct Tail {
  float *x = nullptr;      // [n, W] fp32: the output rows of the residual stream
  void *a = nullptr;       // [n, W] operand format: their attention outputs
  void *h = nullptr;       // [n, W] operand format: LayerNorm output
  void *big = nullptr;     // [n, 4 W] operand format: c_fc output (before that, [n, W]: the output rows' projected queries)
  float *sq = nullptr;     // [n, W] fp32: single-query attention output (text tower; the vision towers use cls_tmp)
};
void plan_tail(Bump &b, Tail &t, int n, int W) {      // sized for fp32 operands (the largest format)
  t.x = (float *)b.take((size_t)n * W * 4);
  t.a = b.take((size_t)n * W * 4);
  t.h = b.take((size_t)n * W * 4);
  t.big = b.take((size_t)n * 4 * W * 4);
  t.sq = (float *)b.take((size_t)n * W * 4);
}
inline bool prune_last(int flags) { return (flags & VTC_TOWER_FULL_LAST_LAYER) == 0; }
// ... and the last block's QUERIES are per-row maps too: only the output rows' queries are
// projected (tail_query: LayerNorm kernel + the Q third of in_proj on n rows), the all-rows projection computes the K and V thirds
// only (ln_proj's column window) and the attention core runs one query per sequence (attention.hip, sq_attn_kernel).
inline bool prune_last_queries(int flags) { return prune_last(flags); }
// the n output rows of the stream -> t.x (fp32, compact); their queries ln_1(x) Wq^T + bq -> t.big [n, W] (operand format)
int tail_query(Fold &f, const vtc_block_w &b, float *x, Tail &t, int n, int W, const int *row_index, int row_mul, int dtype, hipStream_t s) {
  RUN(fold_merge_rows(f, x, n, W, row_index, row_mul, s));
  RUN(launch_gather_rows(x, t.x, n, W * 4, row_index, row_mul, s));
  RUN(launch_layernorm(t.x, b.ln1_g, b.ln1_b, t.h, n, W, dtype, nullptr, 1, false, s));
  return gemm(t.h, b.qkv_w, b.qkv_b, t.big, n, W, W, dtype, VTC_EPI_STORE, dtype, 0, s);
}
// t.x += out_proj(t.a) ; t.x += MLP(ln_2 t.x)      (t.a: the output rows' attention outputs, operand format)
int tail_finish(const vtc_block_w &b, Tail &t, int n, int W, int dtype, hipStream_t s) {
  {
    ProfRegion region(VTC_PROF_REGION_ATTN);
    RUN(gemm(t.a, b.out_w, b.out_b, t.x, n, W, W, dtype, VTC_EPI_RESID, VTC_F32, 0, s));
  }
  Fold nofold;
  return mlp_part(nofold, b, t.x, t.h, t.big, Rows(n), W, dtype, s);
}
// attn_out: [rows, W] operand format, the attention core's output (cls rows already averaged over frames for the TimeSformer);
// the output row of item i is row_index[i] (or i * row_mul) of the stream
int last_block_tail(Fold &f, const vtc_block_w &b, float *x, const void *attn_out, Tail &t, int n, int W, const int *row_index, int row_mul,
                    int dtype, hipStream_t s) {
  {
    ProfRegion region(VTC_PROF_REGION_ATTN);
    RUN(fold_merge_rows(f, x, n, W, row_index, row_mul, s));      // (hi, lo) -> fp32, those rows only
    RUN(launch_gather_rows(x, t.x, n, W * 4, row_index, row_mul, s));
    RUN(launch_gather_rows(attn_out, t.a, n, W * esz(dtype), row_index, row_mul, s));
    RUN(gemm(t.a, b.out_w, b.out_b, t.x, n, W, W, dtype, VTC_EPI_RESID, VTC_F32, 0, s));
  }
  Fold nofold;
  return mlp_part(nofold, b, t.x, t.h, t.big, Rows(n), W, dtype, s);
}

// operand format of text block l: with dtype == VTC_BF16 the first half_layers blocks run on IEEE half
inline int layer_dtype(const vtc_text_w *w, int l, int dtype) {
  return (dtype == VTC_BF16 && l < w->half_layers) ? VTC_F16 : dtype;
}

struct VisionWs {
  float *x, *cls_tmp;
  void *h, *big, *lnp;
  Fold fold;
  Tail tail;
  size_t total;
};

VisionWs plan_vision(const vtc_vision_w *w, int n_items, int F, int dtype, void *ws) {
  Bump b(ws);
  VisionWs v;
  const int P = w->grid * w->grid, T = 1 + P * F, W = w->width;
  const size_t rows = (size_t)pad256(n_items * T);      // padded: the folded-LayerNorm GEMMs run whole 256-row tiles
  const size_t patch_elems = (size_t)n_items * F * P * patch_k_padded(w->patch);
  size_t big_elems = rows * 4 * W;
  if (patch_elems > big_elems) big_elems = patch_elems;
  v.x = (float *)b.take(rows * W * 4);
  v.h = b.take(rows * W * esz(dtype));
  v.big = b.take(big_elems * esz(dtype));
  v.cls_tmp = (float *)b.take((size_t)n_items * (F > P ? F : P) * W * 4);
  v.lnp = b.take((size_t)n_items * W * 4);
  if (dtype != VTC_F32) {
    v.fold.rows_pad = (int)rows;
    v.fold.xb = b.take(rows * W * 2);
    v.fold.xl = b.take(rows * W * 2);
    v.fold.part = (float *)b.take((size_t)(W / 64 + 1) * rows * 8);
    v.fold.stat = (float *)b.take(rows * 8);
    if (rows <= (size_t)SPLITK_MAX_ROWS) v.fold.splitk = (float *)b.take((size_t)SPLITK_MAX * rows * W * 4);
  }
  plan_tail(b, v.tail, n_items, W);
  v.total = b.off;
  return v;
}

struct TextWs {
  float *x;
  void *h, *big, *lnp;
  int *eot;
  int *lens, *offs, *mdev;     // vtc_text_forward2, ragged: per-sequence lengths, their prefix sums, (rows, rows padded to 256)
  Fold fold;
  Tail tail;
  size_t total;
};

TextWs plan_text(int rows_, int n_seq, int W, int dtype, void *ws) {
  Bump b(ws);
  TextWs t;
  const size_t rows = (size_t)pad256(rows_);
  t.x = (float *)b.take(rows * W * 4);
  t.h = b.take(rows * W * esz(dtype));
  t.big = b.take(rows * 4 * W * esz(dtype));
  t.lnp = b.take((size_t)n_seq * W * 4);
  t.eot = (int *)b.take((size_t)n_seq * 4);
  t.lens = (int *)b.take((size_t)n_seq * 4);
  t.offs = (int *)b.take((size_t)(n_seq + 1) * 4);
  t.mdev = (int *)b.take(16);
  if (dtype != VTC_F32) {
    t.fold.rows_pad = (int)rows;
    t.fold.xb = b.take(rows * W * 2);
    t.fold.xl = b.take(rows * W * 2);
    t.fold.part = (float *)b.take((size_t)(W / 64 + 1) * rows * 8);
    t.fold.stat = (float *)b.take(rows * 8);
    if (rows <= (size_t)SPLITK_MAX_ROWS) t.fold.splitk = (float *)b.take((size_t)SPLITK_MAX * rows * W * 4);
  }
  plan_tail(b, t.tail, n_seq, W);
  t.total = b.off;
  return t;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" size_t vtc_vision_workspace_bytes(const vtc_vision_w *w, int n_items, int frames, int dtype) {
  return plan_vision(w, n_items, frames, dtype, nullptr).total;
}

extern "C" int vtc_vision_forward(const vtc_vision_w *w, const void *pixels, int pixel_dtype, int n_items, int F, float *out,
                                  void *ws, size_t ws_bytes, int dtype, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  VTC_CHECK(w && pixels && out && ws, "vision_forward: null argument");
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "vision_forward: bad dtype %d", dtype);
  VTC_CHECK(n_items > 0 && F > 0, "vision_forward: n_items=%d frames=%d", n_items, F);
  const bool tsf = w->nframes > 0;
  VTC_CHECK(tsf || F == 1, "vision_forward: the image tower takes frames == 1 (got %d)", F);
  VTC_CHECK(!tsf || F <= w->nframes, "vision_forward: %d frames > temporal_embed rows %d", F, w->nframes);
  VTC_CHECK(w->width == w->heads * 64, "vision_forward: head_dim must be 64 (width %d, heads %d)", w->width, w->heads);
  const int P = w->grid * w->grid, T = 1 + P * F, W = w->width;
  VTC_CHECK(1 + P <= 272 && F <= 272, "vision_forward: sequence too long (1+P=%d, F=%d; the attention cores cover 272 tokens)", 1 + P, F);
  VTC_CHECK(w->variant == 0 || w->variant == 1, "vision_forward: unknown variant %d", w->variant);
  VTC_CHECK(!(w->variant == 1 && w->nframes == 0), "vision_forward: variant 1 (model/timesformer_clip.py) is a video tower: nframes must be > 0");
  const int rows = n_items * T, res = w->grid * w->patch;
  VisionWs v = plan_vision(w, n_items, F, dtype, ws);
  VTC_CHECK(ws_bytes >= v.total, "vision_forward: workspace too small (%zu < %zu)", ws_bytes, v.total);

  // patch embedding (+pos, +temporal) scattered into the reference's token order, cls rows, ln_pre
  VTC_CHECK(pixel_dtype == VTC_F32 || pixel_dtype == VTC_BF16 || pixel_dtype == VTC_U8 || pixel_dtype == VTC_F16, "vision_forward: bad pixel dtype %d", pixel_dtype);
  {
    GemmEpi e;
    e.mode = EPI_PATCH; e.out_dtype = VTC_F32; e.pos = w->pos; e.temporal = tsf ? w->temporal : nullptr;
    e.P = P; e.F = F; e.T = T; e.ldo = W; e.frames_major = w->variant == 1;
    const void *act = v.big;
    // (few patches: the gather only exists on the 256 x 256 phased kernel, whose single tile would walk K = 3 072 alone -- 113 us
    // at B <= 8 videos; the im2row matrix + 64 x 64 / 128 x 128 tiles take 30-60 us there)
    const bool few = (long)n_items * F * P < 4096;
    if (!few && gemm_patch_gather_supported(n_items * F, w->grid, w->patch, res, pixel_dtype, dtype) && ((size_t)pixels & 15) == 0) {
      // pixels already in the operand format: the GEMM's LDS-DMA reads the patches where they lie (no im2row matrix)
      e.gather = 1; e.grid = w->grid; e.res = res; e.patch = w->patch;
      act = pixels;
    } else if (!few && pixel_dtype == VTC_U8 && gemm_patch_gather_supported(n_items * F, w->grid, w->patch, res, dtype, dtype) && ((size_t)pixels & 15) == 0) {
      // raw uint8 frames (the loader's ToTensor + Normalize, dataset_loaders/dataset_loaders.py:40-49, as the
      // prologue of the patch GEMM): ONE pass turns them into normalised pixels in the operand format -- a quarter of the fp32
      // H2D bytes arrived, no im2row matrix is built -- and the GEMM gathers the patches from that tensor as above
      RUN(launch_pixels_u8_to_operand(pixels, v.big, dtype, n_items * F, res, w->pix_mean, w->pix_std, s));
      e.gather = 1; e.grid = w->grid; e.res = res; e.patch = w->patch;
    } else {
      RUN(launch_im2row(pixels, pixel_dtype, v.big, dtype, n_items * F, w->grid, w->patch, res, w->pix_mean, w->pix_std, s));
    }
    RUN(launch_gemm(act, w->conv_w, nullptr, v.x, n_items * F * P, W, patch_k_padded(w->patch), dtype, e, s));
  }
  RUN(launch_cls_rows(v.x, w->class_embedding, w->pos, n_items, T, W, s));
  Fold &fold = v.fold;
  fold.on = fold_usable(w->blocks, w->layers, W, dtype, tsf, w->flags);
  if (fold.on && w->layers > 0) {
    // ln_pre and the entry into the (hi, lo) stream in one pass (round 6: the LayerNorm kernel + cast_rowstats moved the rows twice)
    RUN(launch_ln_cast_rowstats(v.x, w->ln_pre_g, w->ln_pre_b, fold.xb, fold.xl, fold.stat, rows, W, dtype, s));
    fold.fmt = dtype;
  } else {
    RUN(launch_layernorm(v.x, w->ln_pre_g, w->ln_pre_b, v.x, rows, W, VTC_F32, nullptr, 1, false, s));
  }
  const bool prune = prune_last(w->flags);
  const bool tail_q = prune_last_queries(w->flags) && !(tsf && w->variant == 1);   // ... and its queries (not on the v1 tower's global cls attention)
  for (int l = 0; l < w->layers; ++l) {
    const vtc_block_w &b = w->blocks[l];
    const bool tail = prune && l == w->layers - 1;      // out_proj + MLP of the last block: the cls rows only (last_block_tail)
    const void *tail_src = v.h;                         // where the last attention core's output lies
    {
    ProfRegion region(VTC_PROF_REGION_ATTN);   // time + space branches: what BASELINE.md calls "TimeSformer attention"
    if (tsf && w->variant == 1) {
      // model/timesformer_clip.py:308-315: x += time(ln_time x); x += space(ln_1 x); x += mlp(ln_2 x).
      // Tokens are (frames patches): row item*T + 1 + t*P + n (:392).  In both attentions the cls query
      // attends to every token (:81,:158) -- cls_global_attention -- and a patch query to cls + its
      // same-position frames (time, :161-185) or cls + its same-frame patches (space, :84-108): the
      // generic kernel over [cls, ...] sequences whose own cls output is discarded into cls_tmp.
      RUN(ln_proj(fold, v.x, b.lnt_g, b.lnt_b, b.tqkv_w, b.tqkv_b, b.tqkv_wf, b.tqkv_s, b.tqkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s));
      RUN(launch_attention(v.big, v.h, v.cls_tmp, n_items * P, 1 + F, w->heads, 0, P, 0, T, 0, 1, P, dtype, s));
      RUN(launch_cls_global_attention(v.big, v.h, n_items, T, w->heads, dtype, s));
      RUN(resid_proj(fold, v.h, b.tout_w, b.tout_b, v.x, rows, W, W, dtype, 0, s));
      RUN(ln_proj(fold, v.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s));
      RUN(launch_attention(v.big, v.h, v.cls_tmp, n_items * F, 1 + P, w->heads, 0, F, 0, T, 0, P, 1, dtype, s));
      RUN(launch_cls_global_attention(v.big, v.h, n_items, T, w->heads, dtype, s));
      if (!tail) RUN(resid_proj(fold, v.h, b.out_w, b.out_b, v.x, rows, W, W, dtype, 0, s));
    } else if (tsf) {
      // temporal branch (timesformer_clip_alt.py:142-149): sequences = the F frames of one (item, patch)
      RUN(ln_proj(fold, v.x, b.lnt_g, b.lnt_b, b.tqkv_w, b.tqkv_b, b.tqkv_wf, b.tqkv_s, b.tqkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s));
      RUN(launch_attention(v.big, v.h, nullptr, n_items * P, F, w->heads, 0, P, 1, T, F, 0, 1, dtype, s));
      if (b.tout_w) {
        RUN(gemm(v.h, b.tout_w, b.tout_b, v.big, fold.on ? fold.rows_pad : rows, W, W, dtype, VTC_EPI_STORE, dtype, 0, s));
        RUN(resid_proj(fold, v.big, b.tfc_w, b.tfc_b, v.x, rows, W, W, dtype, T, s));
      } else {  // temporal_fc o out_proj pre-multiplied on the host
        RUN(resid_proj(fold, v.h, b.tfc_w, b.tfc_b, v.x, rows, W, W, dtype, T, s));
      }
      // spatial branch (:152-168): sequences = (item, frame): [cls, the P patches of that frame]
      if (tail && tail_q) {
        // last block: the cls query only (one per item, the same in each of its F sequences), K and V of every row
        RUN(tail_query(fold, b, v.x, v.tail, n_items, W, nullptr, T, dtype, s));
        RUN(ln_proj(fold, v.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s, W, 2 * W));
        RUN(launch_single_query_attention(v.big, v.tail.big, v.cls_tmp, n_items * F, 1 + P, w->heads, F, 0, T, 0, 1, F, nullptr, nullptr, 0, dtype, s));
        RUN(launch_mean_cast(v.cls_tmp, v.tail.a, n_items, F, W, dtype, s));      // cls output = mean over the frames (:166-167)
      } else {
        RUN(ln_proj(fold, v.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s));
        RUN(launch_attention(v.big, v.h, v.cls_tmp, n_items * F, 1 + P, w->heads, 0, F, 0, T, 0, 1, F, dtype, s));
        RUN(launch_cls_mean(v.cls_tmp, v.h, dtype, n_items, F, T, W, s));
        if (!tail) RUN(resid_proj(fold, v.h, b.out_w, b.out_b, v.x, rows, W, W, dtype, 0, s));
      }
    } else if (tail && tail_q) {
      // image tower, last block: the cls query only
      RUN(tail_query(fold, b, v.x, v.tail, n_items, W, nullptr, T, dtype, s));
      RUN(ln_proj(fold, v.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, v.h, v.big, rows, 3 * W, W, dtype, VTC_EPI_STORE, s, W, 2 * W));
      RUN(launch_single_query_attention(v.big, v.tail.big, v.cls_tmp, n_items, T, w->heads, 1, 0, T, 0, 0, 1, nullptr, nullptr, 0, dtype, s));
      RUN(launch_mean_cast(v.cls_tmp, v.tail.a, n_items, 1, W, dtype, s));
    } else {
      RUN(attn_part_contig(fold, b, v.x, v.h, v.big, n_items, T, W, w->heads, 0, dtype, w->flags, s, tail ? &tail_src : nullptr));
    }
    }
    if (tail && tail_q && w->variant == 0) RUN(tail_finish(b, v.tail, n_items, W, dtype, s));
    else if (tail) RUN(last_block_tail(fold, b, v.x, tail_src, v.tail, n_items, W, nullptr, T, dtype, s));
    else RUN(mlp_part(fold, b, v.x, v.h, v.big, rows, W, dtype, s, w->flags));
  }
  // ln_post(x[:,0]) @ proj -- always fp32 (n_items rows only): the embedding the sweep ranks on
  // does not pick up a last bf16 rounding
  if (prune && w->layers > 0) {
    RUN(launch_layernorm(v.tail.x, w->ln_post_g, w->ln_post_b, v.lnp, n_items, W, VTC_F32, nullptr, 1, false, s));
  } else {
    RUN(fold_merge_rows(fold, v.x, n_items, W, nullptr, T, s));
    RUN(launch_layernorm(v.x, w->ln_post_g, w->ln_post_b, v.lnp, n_items, W, VTC_F32, nullptr, T, false, s));
  }
  RUN(gemm(v.lnp, w->proj_t, nullptr, out, n_items, w->embed_dim, W, VTC_F32, VTC_EPI_STORE, VTC_F32, 0, s));
  return 0;
}

// ------------------------------------------------------------------------------------------
extern "C" size_t vtc_text_workspace_bytes(const vtc_text_w *w, int n_seq, int dtype) {
  return plan_text(n_seq * w->ctx, n_seq, w->width, dtype, nullptr).total;
}

namespace {
// The text tower over the sequences of `ids` (two arrays back to back).  mode 0: dense, all ctx positions of every sequence;
// 1: ragged with host-known prefix sums (seq_offsets, total_rows); 2: ragged with NOTHING known to the host -- EOT positions,
// prefix sums and the row count are computed on the device (embed.hip text_prep) and every kernel reads the count there.
// Ragged = only the rows [seq_offsets[s], seq_offsets[s+1]) = tokens 0..EOT of each sequence exist.  Identical outputs: under
// the causal mask no token after EOT can influence the EOT feature, and every other op of the tower is per-row.
int text_forward_impl(const vtc_text_w *w, const TextIds &ids, int mode, const int *seq_offsets, int total_rows, float *out, void *ws,
                      size_t ws_bytes, int dtype, hipStream_t s, const char *who) {
  const int n_seq = ids.n_a + ids.n_b;
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16 || dtype == VTC_F16, "%s: bad dtype %d", who, dtype);
  VTC_CHECK(w->ctx <= 80, "%s: context %d > 80 unsupported", who, w->ctx);
  VTC_CHECK(w->width == w->heads * 64, "%s: head_dim must be 64", who);
  const int W = w->width;
  const int rows_host = mode == 1 ? total_rows : n_seq * w->ctx;          // exact (modes 0, 1) or the dense upper bound (mode 2)
  TextWs t = plan_text(rows_host, n_seq, W, dtype, ws);
  VTC_CHECK(ws_bytes >= t.total, "%s: workspace too small (%zu < %zu)", who, ws_bytes, t.total);
  const int *offs = seq_offsets;
  Rows rows(rows_host);
  if (mode == 2) {
    RUN(launch_text_prep(ids, n_seq, w->ctx, t.lens, t.offs, t.mdev, s));
    offs = t.offs;
    rows = Rows(rows_host, t.mdev);
  }
  if (mode == 0) RUN(launch_text_embed(ids, w->tok_emb, w->pos, t.x, t.eot, n_seq, w->ctx, W, w->vocab, s));
  else RUN(launch_text_embed_ragged(ids, w->tok_emb, w->pos, offs, t.x, t.eot, n_seq, w->ctx, W, w->vocab, s));
  Fold &fold = t.fold;
  fold.on = fold_usable(w->blocks, w->layers, W, dtype, false, w->flags);
  // attention work of the ragged batch for the profiler (the lengths are not known here): rows x (mean length ~ ctx / 2)
  const double attn_flops_per_row = 4.0 * (0.5 * w->ctx) * 64 * w->heads;
  const bool prune = prune_last(w->flags), tail_q = prune_last_queries(w->flags);
  for (int l = 0; l < w->layers; ++l) {
    const vtc_block_w &b = w->blocks[l];
    const int dl = layer_dtype(w, l, dtype);
    const bool tail = prune && l == w->layers - 1;      // out_proj + MLP of the last block: the EOT rows only (last_block_tail)
    const void *tail_src = t.h;
    if (tail && tail_q) {
      // last block: the EOT query only (it sees its whole causal prefix), K and V of every row
      ProfRegion region(VTC_PROF_REGION_ATTN);
      RUN(tail_query(fold, b, t.x, t.tail, n_seq, W, t.eot, 1, dl, s));
      RUN(ln_proj(fold, t.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, t.h, t.big, rows, 3 * W, W, dl, VTC_EPI_STORE, s, W, 2 * W));
      RUN(launch_single_query_attention(t.big, t.tail.big, t.tail.sq, n_seq, 0, w->heads, 1, 0, 0, 0, 0, 1, t.eot, mode == 0 ? nullptr : offs, w->ctx, dl, s));
      RUN(launch_mean_cast(t.tail.sq, t.tail.a, n_seq, 1, W, dl, s));
    } else if (mode == 0) {
      RUN(attn_part_contig(fold, b, t.x, t.h, t.big, n_seq, w->ctx, W, w->heads, 1, dl, w->flags, s, tail ? &tail_src : nullptr));
    } else {
      ProfRegion region(VTC_PROF_REGION_ATTN);
      RUN(ln_proj(fold, t.x, b.ln1_g, b.ln1_b, b.qkv_w, b.qkv_b, b.qkv_wf, b.qkv_s, b.qkv_c, t.h, t.big, rows, 3 * W, W, dl, VTC_EPI_STORE, s));
      RUN(launch_attention_ragged(t.big, t.h, n_seq, w->ctx, w->heads, 1, offs, attn_flops_per_row * (rows.dev ? 1.0 : rows.n), rows.dev, dl, s));
      if (!tail) RUN(resid_proj(fold, t.h, b.out_w, b.out_b, t.x, rows, W, W, dl, 0, s));
    }
    if (tail && tail_q) RUN(tail_finish(b, t.tail, n_seq, W, dl, s));
    else if (tail) RUN(last_block_tail(fold, b, t.x, tail_src, t.tail, n_seq, W, t.eot, 1, dl, s));
    else RUN(mlp_part(fold, b, t.x, t.h, t.big, rows, W, dl, s, w->flags));
  }
  // ln_final on the EOT row only (LayerNorm is per-row, so gathering first is identical), then @ text_projection
  // (always fp32, as for the vision tower)
  if (prune && w->layers > 0) {
    RUN(launch_layernorm(t.tail.x, w->ln_final_g, w->ln_final_b, t.lnp, n_seq, W, VTC_F32, nullptr, 1, false, s));
  } else {
    RUN(fold_merge_rows(fold, t.x, n_seq, W, t.eot, 1, s));
    RUN(launch_layernorm(t.x, w->ln_final_g, w->ln_final_b, t.lnp, n_seq, W, VTC_F32, t.eot, 1, false, s));
  }
  RUN(gemm(t.lnp, w->proj_t, nullptr, out, n_seq, w->embed_dim, W, VTC_F32, VTC_EPI_STORE, VTC_F32, 0, s));
  return 0;
}
}  // namespace

extern "C" int vtc_text_forward(const vtc_text_w *w, const int64_t *ids, int n_seq, float *out, void *ws, size_t ws_bytes,
                                int dtype, void *stream_) {
  VTC_CHECK(w && ids && out && ws, "text_forward: null argument");
  VTC_CHECK(n_seq > 0, "text_forward: n_seq=%d", n_seq);
  return text_forward_impl(w, TextIds{ids, nullptr, n_seq, 0}, 0, nullptr, 0, out, ws, ws_bytes, dtype, (hipStream_t)stream_, "text_forward");
}

extern "C" int vtc_text_forward2(const vtc_text_w *w, const int64_t *ids_a, int n_a, const int64_t *ids_b, int n_b, int ragged,
                                 float *out, void *ws, size_t ws_bytes, int dtype, void *stream_) {
  VTC_CHECK(w && ids_a && out && ws && (ids_b || n_b == 0), "text_forward2: null argument");
  VTC_CHECK(n_a > 0 && n_b >= 0, "text_forward2: n_a=%d n_b=%d", n_a, n_b);
  return text_forward_impl(w, TextIds{ids_a, ids_b, n_a, n_b}, ragged ? 2 : 0, nullptr, 0, out, ws, ws_bytes, dtype, (hipStream_t)stream_,
                           "text_forward2");
}

// Ragged variant with host-known prefix sums: only the rows [seq_offsets[s], seq_offsets[s+1]) are computed (total_rows of them).
extern "C" size_t vtc_text_ragged_workspace_bytes(const vtc_text_w *w, int n_seq, int total_rows, int dtype) {
  return plan_text(total_rows, n_seq, w->width, dtype, nullptr).total;
}

extern "C" int vtc_text_forward_ragged(const vtc_text_w *w, const int64_t *ids, int n_seq, const int *seq_offsets, int total_rows,
                                       float *out, void *ws, size_t ws_bytes, int dtype, void *stream_) {
  VTC_CHECK(w && ids && seq_offsets && out && ws, "text_forward_ragged: null argument");
  VTC_CHECK(n_seq > 0 && total_rows >= n_seq && total_rows <= n_seq * w->ctx, "text_forward_ragged: n_seq=%d total_rows=%d", n_seq, total_rows);
  return text_forward_impl(w, TextIds{ids, nullptr, n_seq, 0}, 1, seq_offsets, total_rows, out, ws, ws_bytes, dtype, (hipStream_t)stream_,
                           "text_forward_ragged");
}

// ------------------------------------------------------------------------------------------
extern "C" size_t vtc_cam_workspace_bytes(const vtc_cam_w *w, int B, int nc, int dtype) {
  TextWs t = plan_text(B * (1 + nc), B, w->width, dtype, nullptr);
  return t.total + align_up((size_t)B * w->width * 4, 256) + align_up(cam_fused_bar_bytes(), 256);
}

extern "C" int vtc_cam_forward(const vtc_cam_w *w, const float *main_feats, const float *comm_feats, const int64_t *comments,
                               int ctx, int B, int nc, float *adapted, void *ws, size_t ws_bytes, int dtype, void *stream_) {
  hipStream_t s = (hipStream_t)stream_;
  VTC_CHECK(w && main_feats && comm_feats && comments && adapted && ws, "cam_forward: null argument");
  VTC_CHECK(dtype == VTC_F32 || dtype == VTC_BF16, "cam_forward: bad dtype %d", dtype);
  VTC_CHECK(B > 0 && nc >= 0 && 1 + nc <= 80, "cam_forward: B=%d nc=%d", B, nc);
  VTC_CHECK(w->heads >= 1 && w->width % w->heads == 0 && (w->width == w->heads * 64 || (w->width / w->heads <= 128 && 1 + nc <= 16)),
            "cam_forward: head_dim %d / %d: 64, or <= 128 with at most 16 tokens per item", w->width, w->heads);
  const int D = w->width, Lc = 1 + nc, rows = B * Lc;
  TextWs t = plan_text(rows, B, D, dtype, ws);
  float *lin = (float *)((char *)ws + t.total);
  VTC_CHECK(ws_bytes >= vtc_cam_workspace_bytes(w, B, nc, dtype), "cam_forward: workspace too small");
  // small batches: the whole module as one cooperative launch (cam.hip)
  if (cam_fused_supported(w, B, nc, dtype)) {
    const int rc = launch_cam_fused(w, main_feats, comm_feats, comments, ctx, B, nc, adapted, t.x, (float *)t.big, (float *)t.h,
                                    (int *)((char *)ws + t.total + align_up((size_t)B * w->width * 4, 256)), s);
    if (rc >= 0) return rc;      // -1: another stream's cooperative launch may still be running: the multi-launch path below
  }
  RUN(launch_cam_tokens(main_feats, comm_feats, comments, w->mask_embedding, t.x, B, nc, ctx, D, s));
  Fold nofold;      // B (1 + nc) tokens: the LayerNorm kernels
  for (int l = 0; l < w->layers; ++l) {
    const vtc_block_w &b = w->blocks[l];
    RUN(attn_part_contig(nofold, b, t.x, t.h, t.big, B, Lc, D, w->heads, 0, dtype, 0, s));
    RUN(mlp_part(nofold, b, t.x, t.h, t.big, rows, D, dtype, s));
  }
  if (!w->init_from_avg) {  // comm_res = final_linear(comm_tfm[0])   model/model.py:161
    RUN(launch_layernorm(t.x, nullptr, nullptr, t.lnp, B, D, dtype, nullptr, Lc, true, s));
    RUN(gemm(t.lnp, w->final_linear, nullptr, lin, B, D, D, dtype, VTC_EPI_STORE, VTC_F32, 0, s));
  }
  RUN(launch_cam_finalize(t.x, lin, main_feats, adapted, B, Lc, D, w->init_from_avg, w->residual_activation, w->squash_scale, w->bn_mean, w->bn_var, s));
  return 0;
}
