"""TEST INFRASTRUCTURE (never imported by the product path): numpy stand-ins for the two halves of the sharded sweep,
include/vtc_hip.h vtc_l2_sweep_shard_rows / vtc_l2_sweep_shard_cols, so that the exchange in vtc_amd/dist.py (slicing,
all-to-all, per-source row bases, padded blocks) can run under gloo on CPU.  They restate the kernels' CONTRACT --
per (column, block of `rb` rows) the three smallest distance keys + the fourth as a bound; a certified candidate set
re-ranked in fp64, brute force otherwise -- not their code.  What they emulate replaces the reference's second
faiss search (model/metric.py:137-146 called from evaluation/eval.py:122-127)."""
import numpy as np

INF_KEY = np.uint32(0x7F800000)


def _keys(dist32, rb):
    """[rows, cols] fp32 distances -> uint32 keys: (bits(max(d, 0)) & ~127) | row-in-block."""
    d = np.maximum(dist32.astype(np.float32), np.float32(0.0))
    bits = d.view(np.uint32) & np.uint32(0xFFFFFF80)
    idx = (np.arange(d.shape[0], dtype=np.uint32) % np.uint32(rb))[:, None]
    return bits | idx


def column_planes(dist32, rb, nblk_pad):
    """planes [4, nblk_pad, cols] (uint32 viewed as int32) of a rank's [rows, cols] distance block."""
    rows, cols = dist32.shape
    k = _keys(dist32, rb)
    out = np.full((4, nblk_pad, cols), INF_KEY, dtype=np.uint32)
    for blk in range(-(-rows // rb)):
        part = np.sort(k[blk * rb:(blk + 1) * rb], axis=0)[:4]
        out[:part.shape[0], blk] = part
    return out.view(np.int32)


def _sqdist(q, g, dtype):
    q, g = q.astype(dtype), g.astype(dtype)
    return (q * q).sum(1)[:, None] + (g * g).sum(1)[None] - 2.0 * (q @ g.T)


def _exact_topk(q, g, depth, cand=None):
    """ids of the `depth` nearest rows of g (restricted to `cand`), fp64 direct differences, ties by lowest index."""
    idx = np.arange(g.shape[0]) if cand is None else np.unique(cand)
    d = ((q.astype(np.float64)[None] - g[idx].astype(np.float64)) ** 2).sum(1)
    order = np.lexsort((idx, d))[:depth]
    return idx[order]


def shard_rows(a_all, b_local, depth, nblk_pad, rb):
    """(ids [n_local, depth] of a_all for each local b row, planes [4, nblk_pad, n_total])."""
    ids = np.stack([_exact_topk(q, a_all, depth) for q in b_local])
    return ids.astype(np.int64), column_planes(_sqdist(b_local, a_all, np.float32), rb, nblk_pad)


def shard_cols(b_all, a_local, depth, planes, src_base, rb, stats=None):
    """planes [n_src, 4, nblk_pad, n_local] int32, src_base [n_src] -> ids [n_local, depth] of b_all for each local a row."""
    planes = np.ascontiguousarray(planes).view(np.uint32)
    n_src, _, nbp, nl = planes.shape
    own = (a_local.astype(np.float32) ** 2).sum(1)
    other_max = (b_all.astype(np.float32) ** 2).sum(1).max()
    kappa = 2.0 ** -16 + 2.0 * a_local.shape[1] * 2.0 ** -24 + 1e-6   # index bits + fp32 accumulation of the stand-in's distances
    base = (np.asarray(src_base, dtype=np.int64)[:, None] + np.arange(nbp, dtype=np.int64)[None] * rb)   # [n_src, nbp]
    out = np.empty((nl, depth), dtype=np.int64)
    brute = 0
    for j in range(nl):
        pool = planes[:, :3, :, j]                                      # [n_src, 3, nbp]
        val = (pool & np.uint32(0xFFFFFF80)).view(np.float32)
        live = pool != INF_KEY
        mins = np.sort(val[:, 0][live[:, 0]])
        theta = (mins[depth - 1] if mins.size >= depth else np.inf) + 2.0 * kappa * (own[j] + other_max)
        take = live & (val <= theta)
        bound = (planes[:, 3, :, j] & np.uint32(0xFFFFFF80)).view(np.float32)
        bound = np.where(planes[:, 3, :, j] == INF_KEY, np.inf, bound).min()
        cand = (base[:, None, :] + (pool & np.uint32(127)).astype(np.int64))[take]
        if bound > theta and cand.size <= 64:
            out[j] = _exact_topk(a_local[j], b_all, depth, cand)
        else:
            brute += 1
            out[j] = _exact_topk(a_local[j], b_all, depth)
    if stats is not None:
        stats["brute"] = brute
    return out


# ---- adversarial case for the certificate's error bound (bf16's unit roundoff is 2^-8, not 2^-9) -----------------
def bf16_round(x):
    """fp32 -> bf16 -> fp32, round to nearest even (what v_cvt_pk_bf16_f32 does on finite values)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(np.float32)


def midpoint_case(n_gallery=2048, d=512, depth=11):
    """(gallery [n, d], query [1, d]) whose coordinates sit on bf16 round-to-even MIDPOINTS so that the operand roundings of a
    plain-bf16 distance are coordinated: the true nearest neighbour X (row 0) is over-estimated by ~ +4.6, the `depth` rows
    Y_t (rows 64 (t + 1): one per block of 64) that are truly 0.3-0.55 FARTHER are under-estimated by ~ -4.6.  The spread
    (9.2) exceeds 2 eps of the halved constant of rounds 1-2 (8.25: X falls outside theta and a 'certified' list misses the
    true #1) and stays inside 2 eps of the correct one (16.5).  Fillers are far away (distance ~ 1000) with norms below |q|.
    Query: first half of the coordinates rounds DOWN (1 + 2^-8 -> 1), second half UP (1 + 3 2^-8 -> 1 + 2^-6)."""
    h = d // 2
    q = np.empty(d, np.float32)
    q[:h] = 1.0 + 2.0 ** -8
    q[h:] = 1.0 + 3 * 2.0 ** -8
    lo_dn = 1.40625 + 2.0 ** -8       # 1 + 52/128 (even mantissa) + half an ulp: rounds down
    lo_up = 1.4140625 + 2.0 ** -8     # 1 + 53/128 (odd mantissa) + half an ulp: rounds up
    rng = np.random.default_rng(7)
    g = rng.choice(np.array([-1.0, 1.0], np.float32), size=(n_gallery, d))
    x = np.full(d, 0.125, np.float32)
    x[:h] = lo_dn
    g[0] = x
    for t in range(depth):
        y = np.full(d, 0.125, np.float32)
        y[h:] = lo_up
        y[:16] = 0.0                                   # + 3.77 on the exact distance
        y[16:16 + 2 * t] = 0.125 - 8 * 2.0 ** -10      # + 0.0276 t (exactly representable in bf16: no rounding of its own)
        g[64 * (t + 1)] = y
    return g, q[None]


def measured_eps(gallery, query, kappa_fp32):
    """Round 4's per-row error bound (sweep.hip: sweep_prep_kernel + minsel_kernel): with x~ = bf16(x) and e = x - x~ the rows'
    ACTUAL rounding errors, | q~.g~ - q.g | = | q~.e_g + e_q.g~ + e_q.e_g | <= |q~||e_g| + |e_q||g~| + |e_q||e_g| (Cauchy-Schwarz),
    twice that on the distance, the gallery side by its maxima, plus kappa_fp32 (|q|^2 + max|g|^2) for the fp32 terms; the same
    1e-4 slack on the measured norms and 1e-3 on the whole as the kernels carry."""
    q, g = query[0].astype(np.float64), gallery.astype(np.float64)
    qt, gt = bf16_round(query[0]).astype(np.float64), bf16_round(gallery).astype(np.float64)
    nq_t, e_q = np.sqrt((qt ** 2).sum()) * 1.0001, np.sqrt(((q - qt) ** 2).sum()) * 1.0001
    ng_t, e_g = np.sqrt((gt ** 2).sum(1)).max() * 1.0001, np.sqrt(((g - gt) ** 2).sum(1)).max() * 1.0001
    return 1.001 * (2.0 * (nq_t * e_g + e_q * ng_t + e_q * e_g) + kappa_fp32 * ((q ** 2).sum() + (g ** 2).sum(1).max()))


def certificate_sets(gallery, query, depth, kappa, bw=64, eps=None):
    """The block-minima certificate of sweep.hip (minsel_kernel) restated for ONE query: plain-bf16 approximate distances in
    fp32, per-block three smallest + the fourth as a bound, u = depth-th smallest block minimum, theta = u + 2 eps with
    eps = kappa (|q|^2 + max|g|^2) (rounds 2-3: a worst-case constant) or the `eps` given (round 4: measured_eps).
    Returns (candidate ids with approx <= theta, certified?, approx distances)."""
    qb, gb = bf16_round(query[0]), bf16_round(gallery)
    qn = np.float32((query[0].astype(np.float64) ** 2).sum())
    gn = (gallery.astype(np.float64) ** 2).sum(1).astype(np.float32)
    dot = (gb.astype(np.float64) @ qb.astype(np.float64)).astype(np.float32)
    approx = np.maximum((qn - np.float32(2.0) * dot) + gn, np.float32(0.0))
    n = gallery.shape[0]
    nblk = -(-n // bw)
    mins, fourth, pool = [], [], []
    for b in range(nblk):
        idx = np.arange(b * bw, min(n, (b + 1) * bw))
        o = idx[np.argsort(approx[idx], kind="stable")]
        mins.append(approx[o[0]])
        pool += list(o[:3])
        fourth.append(approx[o[3]] if len(o) > 3 else np.inf)
    u = np.sort(np.array(mins))[depth - 1]
    theta = u + 2.0 * (kappa * (qn + gn.max()) if eps is None else eps)
    pool = np.array(pool)
    cand = pool[approx[pool] <= theta]
    return cand, bool(min(fourth) > theta and cand.size <= 64), approx
