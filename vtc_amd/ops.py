"""Thin torch-tensor wrappers over the C ABI (include/vtc_hip.h).

torch is used for device memory and the current HIP stream only; every
computation below happens in libvtc_hip.so.  Inputs must live on a ROCm GPU.
"""
from __future__ import annotations

import ctypes as C
import functools
from typing import Optional, Sequence

import torch

from . import _lib as L

_TDT = {torch.float32: L.VTC_F32, torch.bfloat16: L.VTC_BF16, torch.float16: L.VTC_F16}


def _stream() -> int:
    """The current HIP stream of the CURRENT device -- call inside an ``on_device`` wrapper, which makes the
    tensors' device current first."""
    return torch.cuda.current_stream().cuda_stream


def on_device(fn):
    """Every launch goes to the current stream of the device its tensors live on: the wrapper checks that all GPU
    tensor arguments share one device and makes it current for the call (the reference's entry points run on
    ``cuda:<-d>``, evaluation/eval.py:196, without that device being the current one)."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for a in args + tuple(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise RuntimeError(f"vtc_amd.{fn.__name__}: tensors on different devices ({dev} and {a.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _gpu(t: torch.Tensor, dtype=None, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"vtc_amd: {name} must be on the GPU (got {t.device}); this package has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"vtc_amd: {name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def dtype_code(dtype) -> int:
    if isinstance(dtype, int):
        return dtype
    return _TDT[dtype]


def torch_dtype(code: int):
    return {L.VTC_BF16: torch.bfloat16, L.VTC_F16: torch.float16}.get(code, torch.float32)


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---- primitives ---------------------------------------------------------------------------
@on_device
def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = L.EPI_STORE,
         out: Optional[torch.Tensor] = None, out_dtype=None, skip_mod: int = 0) -> torch.Tensor:
    """out[M,N] = epi(a[M,K] @ w[N,K]^T + bias).  a, w: fp32 or bf16 (same dtype)."""
    a, w = _gpu(a, name="a"), _gpu(w, a.dtype, "w")
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    if bias is not None:
        bias = _gpu(bias, torch.float32, "bias")
    if epilogue == L.EPI_RESID:
        assert out is not None and out.dtype == torch.float32 and out.is_contiguous()
    elif out is None:
        out = torch.empty(M, N, dtype=out_dtype or a.dtype, device=a.device)
    L.check(L.lib().vtc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                             M, N, K, _TDT[a.dtype], epilogue, _TDT[out.dtype], skip_mod, _stream()), "vtc_gemm")
    return out


@on_device
def gemm_resid_layernorm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, ln_g: torch.Tensor,
                         ln_b: torch.Tensor, skip_mod: int = 0, ln_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (fp32, in place) += a @ w^T + bias; returns LayerNorm(out) * ln_g + ln_b in a's dtype -- one launch."""
    a, w = _gpu(a, name="a"), _gpu(w, a.dtype, "w")
    M, K = a.shape
    N = w.shape[0]
    assert out.dtype == torch.float32 and out.shape == (M, N) and out.is_contiguous()
    lib = L.lib()
    if not lib.vtc_gemm_resid_layernorm_supported(M, N, K, _TDT[a.dtype]):
        raise ValueError(f"gemm_resid_layernorm: unsupported problem M={M} N={N} K={K} {a.dtype}")
    if ln_out is None:
        ln_out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    ws = workspace(lib.vtc_gemm_resid_layernorm_workspace_bytes(M), a.device)
    L.check(lib.vtc_gemm_resid_layernorm(a.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K,
                                         _TDT[a.dtype], skip_mod, _gpu(ln_g, torch.float32).data_ptr(), _gpu(ln_b, torch.float32).data_ptr(),
                                         ln_out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "vtc_gemm_resid_layernorm")
    return ln_out


@on_device
def layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, out_dtype=torch.float32, rows: Optional[int] = None,
              row_index: Optional[torch.Tensor] = None, row_mul: int = 1) -> torch.Tensor:
    x = _gpu(x, torch.float32, "x")
    width = x.shape[-1]
    n = rows if rows is not None else (row_index.numel() if row_index is not None else x.numel() // width)
    y = torch.empty(n, width, dtype=out_dtype, device=x.device)
    ri = _gpu(row_index, torch.int32, "row_index").data_ptr() if row_index is not None else None
    L.check(L.lib().vtc_layernorm(x.data_ptr(), _gpu(g, torch.float32).data_ptr(), _gpu(b, torch.float32).data_ptr(),
                                  y.data_ptr(), n, width, _TDT[out_dtype], ri, row_mul, _stream()), "vtc_layernorm")
    return y


@on_device
def attention(qkv: torch.Tensor, n_seq: int, L_: int, heads: int, causal: bool = False, s2: int = 1, a0: int = 0,
              a1: Optional[int] = None, a2: int = 0, a3: int = 0, pstride: int = 1, cls_out: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    qkv = _gpu(qkv, name="qkv")
    rows, w3 = qkv.shape
    W = w3 // 3
    assert W == heads * 64
    if a1 is None:
        a1 = L_
    if out is None:
        out = torch.zeros(rows, W, dtype=qkv.dtype, device=qkv.device)
    L.check(L.lib().vtc_attention(qkv.data_ptr(), out.data_ptr(), cls_out.data_ptr() if cls_out is not None else None,
                                  n_seq, L_, heads, int(causal), s2, a0, a1, a2, a3, pstride, _TDT[qkv.dtype], _stream()),
            "vtc_attention")
    return out


@on_device
def single_query_attention(qkv: torch.Tensor, q: torch.Tensor, n_out: int, L_: int, heads: int, s2: int = 1, a0: int = 0,
                           a1: Optional[int] = None, a2: int = 0, a3: int = 0, pstride: int = 1, eot: Optional[torch.Tensor] = None,
                           offs: Optional[torch.Tensor] = None, ctx: int = 0) -> torch.Tensor:
    """One query per sequence over the keys / values of a packed qkv buffer (same row map as ``attention``; or, with ``eot``, the
    text tower's rows base .. eot[o]) -> [n_out, W] fp32.  The last block of a tower: only the output row asks."""
    qkv, q = _gpu(qkv, name="qkv"), _gpu(q, qkv.dtype, "q")
    W = qkv.shape[1] // 3
    assert W == heads * 64 and q.shape[1] == W
    if a1 is None:
        a1 = L_
    out = torch.empty(n_out, W, dtype=torch.float32, device=qkv.device)
    eot_p = _gpu(eot, torch.int32, "eot").data_ptr() if eot is not None else None
    offs_p = _gpu(offs, torch.int32, "offs").data_ptr() if offs is not None else None
    L.check(L.lib().vtc_single_query_attention(qkv.data_ptr(), q.data_ptr(), out.data_ptr(), n_out, L_, heads, s2, a0, a1, a2, a3, pstride,
                                               eot_p, offs_p, ctx, _TDT[qkv.dtype], _stream()), "vtc_single_query_attention")
    return out


# ---- wrapper-level fp32 ops ---------------------------------------------------------------
@on_device
def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    x = _gpu(x, torch.float32, "x")
    out = torch.empty_like(x)
    L.check(L.lib().vtc_normalize_rows(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _stream()), "vtc_normalize_rows")
    return out


@on_device
def mean_groups(x: torch.Tensor, group: int) -> torch.Tensor:
    x = _gpu(x, torch.float32, "x")
    n, d = x.shape
    assert n % group == 0
    out = torch.empty(n // group, d, dtype=torch.float32, device=x.device)
    L.check(L.lib().vtc_mean_groups(x.data_ptr(), out.data_ptr(), n // group, group, d, _stream()), "vtc_mean_groups")
    return out


@on_device
def mean_head_groups(a: torch.Tensor, b: torch.Tensor, group: int) -> torch.Tensor:
    """out[g] = (a[g] + sum_k b[g * group + k]) / (1 + group)   (model/model.py:357-362)."""
    a, b = _gpu(a, torch.float32, "a"), _gpu(b, torch.float32, "b")
    n, d = a.shape
    assert b.shape == (n * group, d)
    out = torch.empty(n, d, dtype=torch.float32, device=a.device)
    L.check(L.lib().vtc_mean_head_groups(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, group, d, _stream()), "vtc_mean_head_groups")
    return out


@on_device
def pack_tokens(tokens: torch.Tensor, offsets: torch.Tensor, ctx: int = 77, sot: int = 49406, eot: int = 49407) -> torch.Tensor:
    """[n_seq, ctx] int64 ids = [sot] + tokens[offsets[s]:offsets[s+1]] + [eot], zero padded; truncated to ctx - 1 ids + eot
    (dataset_loaders/dataset_loaders.py:224-248).  tokens int32 [total], offsets int32 [n_seq + 1], both on the GPU."""
    tokens, offsets = _gpu(tokens, torch.int32, "tokens"), _gpu(offsets, torch.int32, "offsets")
    n = offsets.numel() - 1
    ids = torch.empty(n, ctx, dtype=torch.int64, device=offsets.device)
    L.check(L.lib().vtc_pack_tokens(tokens.data_ptr(), offsets.data_ptr(), n, ctx, sot, eot, ids.data_ptr(), _stream()), "vtc_pack_tokens")
    return ids


@on_device
def segment_mean(x: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    """Mean of rows [offsets[g], offsets[g+1]) per group g (int32 offsets on the GPU)."""
    x = _gpu(x, torch.float32, "x")
    offsets = _gpu(offsets, torch.int32, "offsets")
    n = offsets.numel() - 1
    out = torch.empty(n, x.shape[1], dtype=torch.float32, device=x.device)
    L.check(L.lib().vtc_segment_mean(x.data_ptr(), offsets.data_ptr(), out.data_ptr(), n, x.shape[1], _stream()), "vtc_segment_mean")
    return out


@on_device
def normalize_rows2(x: torch.Tensor, y: torch.Tensor, flag: Optional[torch.Tensor] = None):
    """(x / |x|, y / |y|) row-wise in ONE launch; flag (int32, device) |= 1 / 2 when x / y hold a NaN or inf."""
    x, y = _gpu(x, torch.float32, "x"), _gpu(y, torch.float32, "y")
    assert x.dim() == 2 and y.dim() == 2 and x.shape[1] == y.shape[1]
    ox, oy = torch.empty_like(x), torch.empty_like(y)
    L.check(L.lib().vtc_normalize_rows2(x.data_ptr(), ox.data_ptr(), x.shape[0], y.data_ptr(), oy.data_ptr(), y.shape[0], x.shape[1],
                                        flag.data_ptr() if flag is not None else None, _stream()), "vtc_normalize_rows2")
    return ox, oy


def nonfinite_flag2(a: torch.Tensor, b: torch.Tensor, flag: torch.Tensor) -> torch.Tensor:
    """flag[0] |= 1 when ``a`` holds a NaN / inf, |= 2 when ``b`` does (one launch, no synchronisation).  flag: int32 (or the low word of
    an int64) in device memory."""
    a, b = _gpu(a, torch.float32, "embeddings"), _gpu(b, torch.float32, "embeddings")
    with torch.cuda.device(a.device):
        L.check(L.lib().vtc_nonfinite_flag2(a.data_ptr(), a.numel(), b.data_ptr(), b.numel(), flag.data_ptr(), _stream()), "vtc_nonfinite_flag2")
    return flag


def nonfinite_bits(a: torch.Tensor, b: torch.Tensor) -> int:
    """Synchronous form: the flag word on the host (one 4-byte D2H)."""
    a = _gpu(a, torch.float32, "embeddings")
    return int(nonfinite_flag2(a, b, torch.zeros(1, dtype=torch.int32, device=a.device)).item())


@on_device
def similarity(v: torch.Tensor, t: torch.Tensor, logit_scale: torch.Tensor) -> torch.Tensor:
    v, t = _gpu(v, torch.float32, "v"), _gpu(t, torch.float32, "t")
    ls = _gpu(logit_scale.detach().reshape(1), torch.float32, "logit_scale")
    sim = torch.empty(v.shape[0], t.shape[0], dtype=torch.float32, device=v.device)
    L.check(L.lib().vtc_similarity(v.data_ptr(), t.data_ptr(), v.shape[0], t.shape[0], v.shape[1], ls.data_ptr(),
                                   sim.data_ptr(), _stream()), "vtc_similarity")
    return sim


@on_device
def clip_loss(sim: torch.Tensor) -> torch.Tensor:
    sim = _gpu(sim, torch.float32, "sim")
    n = sim.shape[0]
    assert sim.shape == (n, n)
    ws = workspace(L.lib().vtc_clip_loss_workspace_bytes(n), sim.device)
    loss = torch.empty((), dtype=torch.float32, device=sim.device)
    L.check(L.lib().vtc_clip_loss(sim.data_ptr(), n, loss.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "vtc_clip_loss")
    return loss


# ---- sweep --------------------------------------------------------------------------------
@on_device
def l2_topk(gallery: torch.Tensor, queries: torch.Tensor, depth: int, precision: int = L.SWEEP_EXACT,
            rows_per_block: int = 0, return_dists: bool = True, ws: Optional[torch.Tensor] = None):
    gallery, queries = _gpu(gallery, torch.float32, "gallery"), _gpu(queries, torch.float32, "queries")
    ng, d = gallery.shape
    nq = queries.shape[0]
    need = L.lib().vtc_l2_topk_workspace_bytes(ng, nq, d, precision, rows_per_block)
    if ws is None or ws.numel() < need:
        ws = workspace(need, gallery.device)
    ids = torch.empty(nq, depth, dtype=torch.int64, device=gallery.device)
    dists = torch.empty(nq, depth, dtype=torch.float32, device=gallery.device) if return_dists else None
    L.check(L.lib().vtc_l2_topk(gallery.data_ptr(), queries.data_ptr(), ng, nq, d, depth, precision, rows_per_block,
                                ids.data_ptr(), dists.data_ptr() if dists is not None else None, ws.data_ptr(), ws.numel(),
                                _stream()), "vtc_l2_topk")
    return ids, dists


@on_device
def l2_topk_bidir(a: torch.Tensor, b: torch.Tensor, depth: int, precision: int = L.SWEEP_EXACT, rows_per_block: int = 0,
                  return_dists: bool = True, ws: Optional[torch.Tensor] = None):
    """Both directions from one distance matrix: (ids_b2a [n_b, depth], dists_b2a, ids_a2b [n_a, depth], dists_a2b),
    where ids_b2a = l2_topk(gallery=a, queries=b) and ids_a2b = l2_topk(gallery=b, queries=a)."""
    a, b = _gpu(a, torch.float32, "a"), _gpu(b, torch.float32, "b")
    na, d = a.shape
    nb = b.shape[0]
    need = L.lib().vtc_l2_topk_bidir_workspace_bytes(na, nb, d, precision, rows_per_block)
    if ws is None or ws.numel() < need:
        ws = workspace(need, a.device)
    ids1 = torch.empty(nb, depth, dtype=torch.int64, device=a.device)
    ids2 = torch.empty(na, depth, dtype=torch.int64, device=a.device)
    d1 = torch.empty(nb, depth, dtype=torch.float32, device=a.device) if return_dists else None
    d2 = torch.empty(na, depth, dtype=torch.float32, device=a.device) if return_dists else None
    L.check(L.lib().vtc_l2_topk_bidir(a.data_ptr(), b.data_ptr(), na, nb, d, depth, precision, rows_per_block,
                                      ids1.data_ptr(), d1.data_ptr() if d1 is not None else None,
                                      ids2.data_ptr(), d2.data_ptr() if d2 is not None else None,
                                      ws.data_ptr(), ws.numel(), _stream()), "vtc_l2_topk_bidir")
    return ids1, d1, ids2, d2


def sweep_shard_supported(n_total: int, n_local: int, depth: int) -> bool:
    return bool(L.lib().vtc_l2_sweep_shard_supported(int(n_total), int(n_local), int(depth)))


def sweep_row_block() -> int:
    return int(L.lib().vtc_l2_sweep_row_block())


@on_device
def sweep_shard_rows(a_all: torch.Tensor, b_local: torch.Tensor, depth: int, nblk_pad: int, ws: Optional[torch.Tensor] = None):
    """Rank-local half of the sharded sweep: (ids [n_local, depth] = l2_topk(gallery=a_all, queries=b_local),
    col_planes [4, nblk_pad, n_total] uint32 (as int32 tensor) for the exchange).  include/vtc_hip.h."""
    a_all, b_local = _gpu(a_all, torch.float32, "a_all"), _gpu(b_local, torch.float32, "b_local")
    n, d = a_all.shape
    nl = b_local.shape[0]
    need = L.lib().vtc_l2_sweep_shard_workspace_bytes(n, nl, d)
    if ws is None or ws.numel() < need:
        ws = workspace(need, a_all.device)
    ids = torch.empty(nl, depth, dtype=torch.int64, device=a_all.device)
    planes = torch.empty(4, nblk_pad, n, dtype=torch.int32, device=a_all.device)
    L.check(L.lib().vtc_l2_sweep_shard_rows(a_all.data_ptr(), b_local.data_ptr(), n, nl, d, depth, ids.data_ptr(), None,
                                            planes.data_ptr(), nblk_pad, ws.data_ptr(), ws.numel(), _stream()),
            "vtc_l2_sweep_shard_rows")
    return ids, planes


@on_device
def sweep_shard_cols(b_all: torch.Tensor, a_local: torch.Tensor, depth: int, planes: torch.Tensor, src_base: torch.Tensor,
                     ws: Optional[torch.Tensor] = None):
    """Second half, after the exchange: planes [n_src, 4, nblk_pad, n_local] int32, src_base [n_src] int32 ->
    ids [n_local, depth] = l2_topk(gallery=b_all, queries=a_local)."""
    b_all, a_local = _gpu(b_all, torch.float32, "b_all"), _gpu(a_local, torch.float32, "a_local")
    planes, src_base = _gpu(planes, torch.int32, "planes"), _gpu(src_base, torch.int32, "src_base")
    n, d = b_all.shape
    nl = a_local.shape[0]
    n_src, four, nblk_pad, nl2 = planes.shape
    if four != 4 or nl2 != nl or src_base.numel() != n_src:
        raise ValueError(f"sweep_shard_cols: planes {tuple(planes.shape)} / src_base {tuple(src_base.shape)} do not match n_local={nl}")
    need = L.lib().vtc_l2_sweep_shard_workspace_bytes(n, nl, d)
    if ws is None or ws.numel() < need:
        ws = workspace(need, b_all.device)
    ids = torch.empty(nl, depth, dtype=torch.int64, device=b_all.device)
    L.check(L.lib().vtc_l2_sweep_shard_cols(b_all.data_ptr(), a_local.data_ptr(), n, nl, d, depth, planes.data_ptr(), n_src,
                                            nblk_pad, src_base.data_ptr(), ids.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                            _stream()), "vtc_l2_sweep_shard_cols")
    return ids


def recall_bidir_supported(n: int, d: int) -> bool:
    return bool(L.lib().vtc_l2_recall_bidir_supported(int(n), int(d)))


@on_device
def recall_bidir(a: torch.Tensor, b: torch.Tensor, k_vals: Sequence[int], ws: Optional[torch.Tensor] = None,
                 hits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """R@K hit counters of both directions of n paired rows WITHOUT the sorted neighbour lists (vtc_l2_recall_bidir: one distance GEMM +
    one rank launch): hits[0, j] += #{i : a_i among the k_j nearest a's of b_i} (= RecallAtK.compute(a, b) x n), hits[1, j] the transposed
    direction (= compute(b, a) x n).  The same counters as l2_topk_bidir(depth = max k + 1, EXACT) + recall_hits_pair.
    Non-finite rows are misses and raise bit 40 of hits[:, 0] (L.RECALL_NONFINITE): split_recall_counters() on the host copy."""
    a, b = _gpu(a, torch.float32, "a"), _gpu(b, torch.float32, "b")
    n, d = a.shape
    assert b.shape == (n, d) and 1 <= len(k_vals) <= 4
    if hits is None:
        hits = torch.zeros(2, len(k_vals), dtype=torch.int64, device=a.device)
    assert hits.shape == (2, len(k_vals)) and hits.is_contiguous() and hits.dtype == torch.int64
    need = L.lib().vtc_l2_recall_bidir_workspace_bytes(n, d)
    if ws is None or ws.numel() < need or ws.device != a.device:
        ws = workspace(need, a.device)
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    L.check(L.lib().vtc_l2_recall_bidir(a.data_ptr(), b.data_ptr(), n, d, ks, len(k_vals), hits[0].data_ptr(), hits[1].data_ptr(),
                                        ws.data_ptr(), ws.numel(), _stream()), "vtc_l2_recall_bidir")
    return hits


def split_recall_counters(hits_host: torch.Tensor):
    """(counters, nonfinite) of a HOST copy of the recall-only sweeps' hit counters: the NaN / inf marker (bit 40 and above of the first
    counter of a direction, summed over ranks) taken off."""
    h = hits_host.numpy()                # (numpy: a torch CPU op costs microseconds that the 10k sweep's 0.28 ms would show)
    return torch.from_numpy(h & (L.RECALL_NONFINITE - 1)), bool((h >> 40).any())


def recall_planes(k_vals: Sequence[int], n_total: int) -> int:
    """Key planes per (column, block) that the recall-only sweep keeps for these k at this gallery size (vtc_l2_recall_planes): 2, 3 or 4."""
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    return int(L.lib().vtc_l2_recall_planes(ks, len(k_vals), int(n_total)))


def recall_shard_supported(n_total: int, n_local: int, d: int) -> bool:
    return bool(L.lib().vtc_l2_recall_shard_supported(int(n_total), int(n_local), int(d)))


@on_device
def recall_shard_rows(a_all: torch.Tensor, b_local: torch.Tensor, row_base: int, k_vals: Sequence[int], nblk_pad: int, hits: torch.Tensor,
                      ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rank-local half of the sharded sweep with the recall-only finish (vtc_l2_recall_shard_rows): hits [nk] int64 += this rank's
    counters of RecallAtK.compute(a, b); returns col_planes [P, nblk_pad, n_total] with P = vtc_l2_recall_planes(k_vals, n_local) = 2, 3 or 4 (int32 tensor) for the exchange."""
    a_all, b_local = _gpu(a_all, torch.float32, "a_all"), _gpu(b_local, torch.float32, "b_local")
    n, d = a_all.shape
    nl = b_local.shape[0]
    assert hits.shape == (len(k_vals),) and hits.dtype == torch.int64 and hits.is_contiguous() and hits.device == a_all.device
    need = L.lib().vtc_l2_sweep_shard_workspace_bytes(n, nl, d)
    if ws is None or ws.numel() < need or ws.device != a_all.device:
        ws = workspace(need, a_all.device)
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    planes = torch.empty(L.lib().vtc_l2_recall_planes(ks, len(k_vals), n), nblk_pad, n, dtype=torch.int32, device=a_all.device)
    L.check(L.lib().vtc_l2_recall_shard_rows(a_all.data_ptr(), b_local.data_ptr(), n, nl, int(row_base), d, ks, len(k_vals), hits.data_ptr(),
                                             planes.data_ptr(), nblk_pad, ws.data_ptr(), ws.numel(), _stream()), "vtc_l2_recall_shard_rows")
    return planes


@on_device
def recall_shard_cols(b_all: torch.Tensor, a_local: torch.Tensor, row_base: int, k_vals: Sequence[int], planes: torch.Tensor,
                      src_bounds: torch.Tensor, hits: torch.Tensor, ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Second half, after the exchange: planes [n_src, 4, nblk_pad, n_local] int32, src_bounds [n_src + 1] int32 (device); hits [nk] int64
    += this rank's counters of RecallAtK.compute(b, a)."""
    b_all, a_local = _gpu(b_all, torch.float32, "b_all"), _gpu(a_local, torch.float32, "a_local")
    planes, src_bounds = _gpu(planes, torch.int32, "planes"), _gpu(src_bounds, torch.int32, "src_bounds")
    n, d = b_all.shape
    nl = a_local.shape[0]
    n_src, npl, nblk_pad, nl2 = planes.shape
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    if npl != L.lib().vtc_l2_recall_planes(ks, len(k_vals), n) or nl2 != nl or src_bounds.numel() != n_src + 1:
        raise ValueError(f"recall_shard_cols: planes {tuple(planes.shape)} / src_bounds {tuple(src_bounds.shape)} do not match n_local={nl}, "
                         f"k_vals={list(k_vals)}")
    assert hits.shape == (len(k_vals),) and hits.dtype == torch.int64 and hits.is_contiguous() and hits.device == b_all.device
    need = L.lib().vtc_l2_sweep_shard_workspace_bytes(n, nl, d)
    if ws is None or ws.numel() < need or ws.device != b_all.device:
        ws = workspace(need, b_all.device)
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    L.check(L.lib().vtc_l2_recall_shard_cols(b_all.data_ptr(), a_local.data_ptr(), n, nl, int(row_base), d, ks, len(k_vals), planes.data_ptr(),
                                             n_src, nblk_pad, src_bounds.data_ptr(), hits.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
            "vtc_l2_recall_shard_cols")
    return hits


@on_device
def recall_hits_pair(ids_a: torch.Tensor, ids_b: torch.Tensor, k_vals: Sequence[int], target_offset: int, hits: torch.Tensor) -> torch.Tensor:
    """hits[0] += hits of ids_a, hits[1] += hits of ids_b (both [n, depth], same targets): one launch for both directions."""
    ids_a, ids_b = _gpu(ids_a, torch.int64, "ids_a"), _gpu(ids_b, torch.int64, "ids_b")
    assert ids_a.shape == ids_b.shape and hits.shape[0] == 2 and hits.is_contiguous()
    nq, depth = ids_a.shape
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    L.check(L.lib().vtc_recall_hits_pair(ids_a.data_ptr(), ids_b.data_ptr(), nq, depth, int(target_offset), ks, len(k_vals),
                                         hits[0].data_ptr(), hits[1].data_ptr(), _stream()), "vtc_recall_hits_pair")
    return hits


@on_device
def recall_hits(ids: torch.Tensor, k_vals: Sequence[int], target_offset: int = 0,
                hits: Optional[torch.Tensor] = None) -> torch.Tensor:
    ids = _gpu(ids, torch.int64, "ids")
    nq, depth = ids.shape
    if hits is None:
        hits = torch.zeros(len(k_vals), dtype=torch.int64, device=ids.device)
    ks = (C.c_int * len(k_vals))(*[int(k) for k in k_vals])
    L.check(L.lib().vtc_recall_hits(ids.data_ptr(), nq, depth, int(target_offset), ks, len(k_vals), hits.data_ptr(), _stream()),
            "vtc_recall_hits")
    return hits
