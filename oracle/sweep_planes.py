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
