"""One-process-per-GPU scale-out of the eval path (torch.distributed; backend "nccl" = RCCL).

The reference is single-process (nn.DataParallel, train.py:77-80; eval is single-GPU,
evaluation/eval.py:196).  Here encoding is embarrassingly parallel over pairs -- each rank
encodes its shard, no collective -- and the N x N sweep has exactly one exchange step
(SURVEY 8e):

    all_gather(V_r), all_gather(T_r)          [N/G, 512] fp32 per rank (2.56 MB at 10k, 12.8 MB at 50k)
    rank r sweeps its own query rows against the full gallery, both directions
    all_reduce(sum) of the 2 x len(k) int64 hit counters

No cross-rank top-k merge is needed there because a rank owns whole query rows -- at the price of two
[N/G, N] distance GEMMs per rank.  With VTC_SWEEP_EXACT the ranks instead run ONE GEMM each (rank r: its
b rows x all a columns): the row direction is complete locally, and for the column direction every rank
keeps, per column and per block of 128 of its rows, the three smallest distance keys + a bound
(include/vtc_hip.h, vtc_l2_sweep_shard_rows), sends each column owner its slice

    all_to_all(col_planes[:, :, lo_s:hi_s])   4 x ceil(N/G/128) x N/G uint32 per rank pair (4.9 MB at 50k, G=8)

and the owner certifies + re-ranks in fp64 across all sources (vtc_l2_sweep_shard_cols): the same ids as the
single-GPU search, bit for bit.  Works with "gloo" on CPU tensors for the collectives' bookkeeping
(tests inject CPU stand-ins for the two kernels), the sweep itself is HIP-only.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); single process otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs: several ranks on ONE card (VTC_LOCAL_DEVICE=0) over gloo (VTC_DIST_BACKEND=gloo)
    local = int(os.environ.get("VTC_LOCAL_DEVICE", local))
    backend = os.environ.get("VTC_DIST_BACKEND", backend)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)     # every backend: the rank's tensors and launches live on its own card
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1 and dist.is_initialized():
        mark_shared_cards(local)
    return rank, local, world


def card_identity(local: int):
    """(host, card) of the device this rank computes on -- the card by its UUID / PCI bus id, which survives HIP_VISIBLE_DEVICES
    renumbering and LOCAL_RANK modulo games; a CPU-only rank (gloo tests) is its own card."""
    import socket
    host = socket.gethostname()
    if not torch.cuda.is_available():
        return (host, f"cpu-{os.getpid()}")
    pr = torch.cuda.get_device_properties(local)
    card = str(getattr(pr, "uuid", "")) or ""
    if not card or set(card) <= set("0-"):
        card = f"pci-{getattr(pr, 'pci_domain_id', 0)}:{getattr(pr, 'pci_bus_id', local)}:{getattr(pr, 'pci_device_id', 0)}"
    return (host, card)


def mark_shared_cards(local: int, identities=None) -> bool:
    """ranks that SHARE a card (however they came to: VTC_LOCAL_DEVICE, LOCAL_RANK modulo the device count, a narrowed
    HIP_VISIBLE_DEVICES) must not take the one-launch CAM -- its software grid barrier needs the whole grid resident at once, which
    an ordinary launch can only promise against the process's OWN kernels; with another process's kernels on the same CUs the
    barrier gives up (NaN embeddings, vtc_amd/csrc/cam.hip).  Detected directly: one all_gather of (host, card id) after
    init_process_group; the result travels as the per-model flag VTC_CAM_NO_FUSED (towers.PackedCam), not as an environment
    variable a library static may already have read.  One process per GPU -- the production layout -- is not affected."""
    from . import towers
    mine = card_identity(local)
    if identities is None:
        identities = [None] * dist.get_world_size()
        dist.all_gather_object(identities, mine)
    shared = sum(1 for i in identities if tuple(i) == tuple(mine)) > 1
    towers.set_cam_shared_card(shared)
    return shared


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous shards, sizes differ by at most one: rank r owns [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_rows(x: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """Concatenate every rank's [n_r, D] rows in rank order (shards may differ by one row)."""
    if world == 1:
        return x
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = x
    if x.shape[0] < mx:
        pad = torch.cat([x, x.new_zeros(mx - x.shape[0], x.shape[1])])
    out = x.new_empty(world * mx, x.shape[1])
    dist.all_gather_into_tensor(out, pad.contiguous())
    if all(hi - lo == mx for lo, hi in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


RANK_PATH = os.environ.get("VTC_SWEEP_RANK", "1") != "0"      # world 1, EXACT: hit counters without sorted lists (as host/metric.py RecallAtK.rank_path)
RANK_MIN_ROWS = 1024
BIDIR_MIN_ROWS = 5120       # as host/metric.py RecallAtK.bidir_min_rows (tools/bidir_threshold.py: EXACT 6k 0.49 vs 0.6+, 10k 0.64 vs 0.79 ms)
BIDIR_MIN_ROWS_F32 = 3000


def one_matrix_sharded(n_total: int, precision: int, world: int, depth: int) -> bool:
    """True when world > 1 ranks take the one-GEMM-per-rank path (VTC_SWEEP_EXACT, shapes the block-minima sweep covers;
    VTC_SWEEP_SHARD_TWO=1 forces the two-search path)."""
    if world <= 1 or precision != 3 or os.environ.get("VTC_SWEEP_SHARD_TWO") == "1":
        return False
    from . import ops
    return all(ops.sweep_shard_supported(n_total, hi - lo, depth)
               for lo, hi in (shard_bounds(n_total, r, world) for r in range(world)))


def rank_sharded(n_total: int, d: int, precision: int, world: int, nk: int = 3) -> bool:
    """True when world > 1 ranks take the one-GEMM-per-rank path with the recall-only finish (hit counters from the ranks of the paired
    rows: no sorted lists; VTC_SWEEP_RANK=0 or VTC_SWEEP_SHARD_TWO=1 turn it off)."""
    if world <= 1 or precision != 3 or not RANK_PATH or nk > 4 or os.environ.get("VTC_SWEEP_SHARD_TWO") == "1":
        return False
    from . import ops
    return all(ops.recall_shard_supported(n_total, hi - lo, d) for lo, hi in (shard_bounds(n_total, r, world) for r in range(world)))


def sweep_path(n_total: int, precision: int, world: int, depth: int = 11, d: int = 512) -> str:
    if world > 1 and rank_sharded(n_total, d, precision, world):
        return "one [N/G, N] distance GEMM per rank, column block minima exchanged (all-to-all), ranks of the paired rows (no sorted lists)"
    if world > 1:
        return ("one [N/G, N] distance GEMM per rank, column block minima exchanged (all-to-all)"
                if one_matrix_sharded(n_total, precision, world, depth) else "two searches per rank ([N/G, N] blocks)")
    if precision == 3 and RANK_PATH and n_total >= RANK_MIN_ROWS:
        return "one distance matrix, ranks of the paired rows (no sorted lists)"
    one = n_total >= (BIDIR_MIN_ROWS_F32 if precision == 0 else BIDIR_MIN_ROWS)
    return "one distance matrix, row + column top-k" if one else "two searches per rank ([N/G, N] blocks)"


_A2A_SEND: dict = {}      # exchange_column_planes' padded send buffer / pinned host staging buffers, by shape
A2A_MODE = os.environ.get("VTC_A2A", "single")     # "single": one all_to_all_single on a padded [G, 4, nblk, max shard] buffer; "list": all_to_all on per-rank slices


def exchange_column_planes(planes: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """planes [P, nblk_pad, n_total] of this rank's rows (P = 2, 3 or 4 key planes: vtc_l2_recall_planes) -> [world, P, nblk_pad, n_local]: what every rank (in rank order)
    holds for THIS rank's columns.  One equal-split all_to_all_single (shards differ by at most one column: the buffer is
    padded to the largest) -- the collective form RCCL and gloo both implement; gloo (tests, one-card rehearsals) has no
    device-memory transport, so there the buffer is staged through host memory."""
    bounds = [shard_bounds(n_total, r, world) for r in range(world)]
    lo, hi = bounds[rank]
    mx = max(b - a for a, b in bounds)
    P, NB = planes.shape[0], planes.shape[1]
    if planes.dim() != 3 or planes.shape[2] != n_total or not planes.is_contiguous():
        raise ValueError(f"exchange_column_planes: planes must be a contiguous [P, nblk_pad, n_total = {n_total}] tensor, got "
                         f"{tuple(planes.shape)} (contiguous: {planes.is_contiguous()})")
    nccl = dist.get_backend() == "nccl"
    if A2A_MODE == "list" and nccl:
        send = [planes[:, :, a:b].contiguous() for a, b in bounds]
        recv = planes.new_empty(world, P, NB, hi - lo)
        dist.all_to_all(list(recv.unbind(0)), send)
        return recv
    if all(b - a == mx for a, b in bounds):
        send = planes.reshape(P, NB, world, mx).permute(2, 0, 1, 3).contiguous()
    else:
        # ragged shards: a padded send buffer, kept per (shape, dtype, device) -- the padding columns are sliced off on receipt, so
        # they need no zeroing and the buffer no re-allocation per call
        key = (world, P, NB, mx, planes.dtype, planes.device)
        send = _A2A_SEND.get(key)
        if send is None:
            _A2A_SEND.clear()
            send = _A2A_SEND[key] = planes.new_zeros(world, P, NB, mx)
        for r, (a, b) in enumerate(bounds):
            send[r, :, :, : b - a].copy_(planes[:, :, a:b])
    if nccl:
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
    else:
        # gloo (tests, one-card rehearsals): staged through host memory -- pinned and cached when the planes live on a GPU
        if send.is_cuda:
            hk = ("host",) + tuple(send.shape) + (send.dtype,)
            bufs = _A2A_SEND.get(hk)
            if bufs is None:
                for k_ in [k_ for k_ in _A2A_SEND if k_ and k_[0] == "host"]:      # one pinned pair at a time: the last shape
                    del _A2A_SEND[k_]
                bufs = _A2A_SEND[hk] = (torch.empty(send.shape, dtype=send.dtype, pin_memory=True), torch.empty(send.shape, dtype=send.dtype, pin_memory=True))
            host, got = bufs
            host.copy_(send)
        else:
            host, got = send, torch.empty_like(send)
        dist.all_to_all_single(got, host)
        recv = got.to(planes.device)
    return recv if hi - lo == mx else recv[..., : hi - lo].contiguous()


def sweep_workspace_bytes(n_total: int, n_local: int, d: int, precision: int, world: int) -> int:
    """Bytes of caller-owned workspace that sharded_recall needs for this shape (largest of the paths it may take)."""
    from . import _lib as L
    lib = L.lib()
    need = lib.vtc_l2_topk_workspace_bytes(n_total, n_local, d, precision, 0)
    if world == 1:
        need = max(need, lib.vtc_l2_topk_bidir_workspace_bytes(n_total, n_total, d, precision, 0))
    elif precision == 3:
        need = max(need, lib.vtc_l2_sweep_shard_workspace_bytes(n_total, n_local, d))
    return int(need)


class _PhaseClock:
    """HIP events on the launch stream between the steps of a sweep: `phases[name + "_ms"]` += the GPU time since the previous mark."""

    def __init__(self, phases: Optional[dict], on_gpu: bool):
        self.phases, self.on, self.marks = phases, phases is not None and on_gpu, []

    def mark(self, name: str):
        if self.on:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def close(self, path: str, exchange: Optional[str] = None):
        """Call after the final D2H (the events have completed)."""
        if self.phases is None:
            return
        for (_, e0), (name, e1) in zip(self.marks, self.marks[1:]):
            self.phases[name + "_ms"] = round(self.phases.get(name + "_ms", 0.0) + e0.elapsed_time(e1), 4)
        self.phases["path"] = path
        if exchange is not None:
            self.phases["exchange"] = exchange


def _exchange_name() -> str:
    return (f"all_to_all{'' if A2A_MODE == 'list' else '_single'} (RCCL)" if dist.get_backend() == "nccl"
            else f"all_to_all_single through host memory ({dist.get_backend()})")


def gather_both(feats_a_local: torch.Tensor, feats_b_local: torch.Tensor, n_total: int, rank: int, world: int):
    """(a_all, b_all): every rank's rows in rank order.  ONE exchange for both embedding sets ([n_r, 2D] rows) when they have the same
    width: at 10k x 512 the all-gather is latency-bound, a second collective costs as much as the first."""
    if world > 1 and feats_a_local.shape[1] == feats_b_local.shape[1] and feats_a_local.dtype == feats_b_local.dtype:
        d = feats_a_local.shape[1]
        ab = all_gather_rows(torch.cat([feats_a_local, feats_b_local], dim=1), n_total, rank, world)
        return ab[:, :d].contiguous(), ab[:, d:].contiguous()
    return all_gather_rows(feats_a_local, n_total, rank, world), all_gather_rows(feats_b_local, n_total, rank, world)


def _flag_nonfinite(flag: torch.Tensor, a: torch.Tensor, b: torch.Tensor):
    """flag (one int64 slot behind the hit counters, so it travels with their all-reduce and D2H) |= 1 / 2 when a / b hold a NaN or inf."""
    if a.is_cuda:
        from . import ops
        ops.nonfinite_flag2(a, b, flag)
    else:
        flag[0] = int(not bool(torch.isfinite(a).all())) | 2 * int(not bool(torch.isfinite(b).all()))


# ---- the five paths of sharded_recall: each fills `hits` [2, len(ks)] (this rank's partial counters) or returns the sorted ids -------------
def _world1_rank_path(a_all, b_all, ks, hits, ws, clock):
    """One rank owns every pair, EXACT: the hit counters come straight from the distance GEMM's key planes (vtc_l2_recall_bidir: the
    rank of each query's own gallery row; no sorted lists) -- what RecallAtK.compute_both does on one GPU."""
    from . import ops
    ops.recall_bidir(a_all, b_all, ks, ws=ws, hits=hits)
    clock.mark("bidir_gemm_rank")


def _world1_one_matrix(a_all, b_all, depth, precision, ws, clock):
    """One rank owns the whole matrix: the sorted ids of both directions from one distance GEMM (vtc_l2_topk_bidir)."""
    from . import ops
    i1, _, i2, _ = ops.l2_topk_bidir(a_all, b_all, depth, precision=precision, return_dists=False, ws=ws)
    clock.mark("bidir_gemm_select")
    return i1, i2


def _rank_sharded(a_all, b_all, a_local, b_local, n_total, lo, ks, hits, rank, world, rank_ops, ws, clock):
    """One [N/G, N] GEMM per rank, recall-only finish: this rank's hit counters of the row direction come with the GEMM, the column
    block minima go to the column owners (one all-to-all), whose rank launch counts the other direction."""
    if rank_ops is None:
        from . import ops
        rank_ops = (lambda a_, b_, base, ks_, nbp, h: ops.recall_shard_rows(a_, b_, base, ks_, nbp, h, ws=ws),
                    lambda b_, a_, base, ks_, pl, sb, h: ops.recall_shard_cols(b_, a_, base, ks_, pl, sb, h, ws=ws), ops.sweep_row_block())
    rows_fn, cols_fn, rb = rank_ops
    bounds = [shard_bounds(n_total, r, world) for r in range(world)]
    nblk_pad = -(-max(h - l for l, h in bounds) // rb)
    planes = rows_fn(a_all, b_local, lo, ks, nblk_pad, hits[0])
    clock.mark("rows_gemm_rank")
    recv = exchange_column_planes(planes, n_total, rank, world)
    clock.mark("alltoall")
    src_bounds = torch.tensor([l for l, _ in bounds] + [n_total], dtype=torch.int32, device=planes.device)
    cols_fn(b_all, a_local, lo, ks, recv, src_bounds, hits[1])
    clock.mark("cols_rank")


def _one_matrix_sharded(a_all, b_all, a_local, b_local, n_total, depth, rank, world, shard_ops, ws, clock):
    """One [N/G, N] GEMM per rank, sorted ids: rows finished locally, column block minima to the column owners, who certify + re-rank."""
    if shard_ops is None:
        from . import ops
        shard_ops = (lambda a_, b_, d_, nbp: ops.sweep_shard_rows(a_, b_, d_, nbp, ws=ws),
                     lambda b_, a_, d_, pl, sb: ops.sweep_shard_cols(b_, a_, d_, pl, sb, ws=ws), ops.sweep_row_block())
    rows_fn, cols_fn, rb = shard_ops
    bounds = [shard_bounds(n_total, r, world) for r in range(world)]
    nblk_pad = -(-max(h - l for l, h in bounds) // rb)
    i1, planes = rows_fn(a_all, b_local, depth, nblk_pad)          # gallery a, queries b: this rank's b rows
    clock.mark("rows_gemm_select")
    recv = exchange_column_planes(planes, n_total, rank, world)
    clock.mark("alltoall")
    src_base = torch.tensor([l for l, _ in bounds], dtype=torch.int32, device=planes.device)
    i2 = cols_fn(b_all, a_local, depth, recv, src_base)            # gallery b, queries a: this rank's a rows
    clock.mark("cols_select")
    return i1, i2


def _two_searches(a_all, b_all, a_local, b_local, depth, topk, clock):
    """compute(a, b): gallery a, queries b (model/metric.py:137-146) and the transposed search; this rank owns query rows [lo, hi)."""
    i1 = topk(a_all, b_local, depth)
    clock.mark("search_b_from_a")
    i2 = topk(b_all, a_local, depth)
    clock.mark("search_a_from_b")
    return i1, i2


def _hits_from_ids(both, ks, lo, hi, hits):
    """hits[d, j] = #{ local query i : lo + i among the first ks[j] ids of its row } for the two id arrays."""
    if both[0].is_cuda and len(ks) <= 4 and both[0].shape == both[1].shape:
        from . import ops
        ops.recall_hits_pair(both[0], both[1], ks, lo, hits)        # both directions: one launch
        return
    for d_, ids in enumerate(both):
        if ids.is_cuda and len(ks) <= 4:
            from . import ops
            ops.recall_hits(ids, ks, target_offset=lo, hits=hits[d_])
        else:
            tgt = torch.arange(lo, hi, device=ids.device)[:, None]
            for j, k in enumerate(ks):
                hits[d_, j] = (ids[:, :k] == tgt).any(dim=1).sum()


def sharded_recall(feats_a_local: torch.Tensor, feats_b_local: torch.Tensor, n_total: int, k_vals: Sequence[int],
                   rank: int, world: int,
                   topk: Optional[Callable[[torch.Tensor, torch.Tensor, int], torch.Tensor]] = None,
                   precision: int = 3,    # _lib.SWEEP_EXACT
                   ws: Optional[torch.Tensor] = None,
                   shard_ops: Optional[tuple] = None,
                   phases: Optional[dict] = None,
                   rank_ops: Optional[tuple] = None):
    """R@K both directions for row-sharded embeddings.

    Returns ({k: recall b_from_a-direction as RecallAtK.compute(a, b)}, {k: compute(b, a)}).
    ``topk(gallery, queries, depth) -> ids`` defaults to the HIP sweep; tests inject a CPU one to exercise the sharding logic under
    gloo.  ``ws``: a caller-owned uint8 workspace reused across calls (grown by the ops layer when too small).
    ``shard_ops = (rows_fn, cols_fn, row_block)``: stand-ins for ops.sweep_shard_rows / ops.sweep_shard_cols (tests: the
    one-GEMM-per-rank exchange under gloo).  ``rank_ops = (rows_fn, cols_fn, row_block)``: stand-ins for ops.recall_shard_rows /
    ops.recall_shard_cols (the same exchange with the recall-only finish: hit counters, no ids).
    ``phases``: a dict that receives this rank's GPU time per phase in ms (HIP events on the launch stream: allgather /
    rows_gemm_select / alltoall / cols_select / search_a / search_b / hits / allreduce) and the path taken -- what a scaling run is
    read from.

    The dispatcher only: one function per path above (world 1: rank path, one matrix; world > 1: rank-sharded, one-matrix sharded;
    any world: two searches).  Non-finite embeddings on ANY rank raise ValueError on EVERY rank (the flag word rides behind the hit
    counters through their all-reduce)."""
    lo, hi = shard_bounds(n_total, rank, world)
    clock = _PhaseClock(phases, feats_a_local.is_cuda)
    clock.mark("start")
    assert feats_a_local.shape[0] == hi - lo and feats_b_local.shape[0] == hi - lo
    a_all, b_all = gather_both(feats_a_local, feats_b_local, n_total, rank, world)
    clock.mark("allgather")
    depth = min(int(max(k_vals)) + 1, n_total)
    ks = [min(int(k), depth) for k in k_vals]
    d = feats_a_local.shape[1]
    same_width = d == feats_b_local.shape[1]
    acc = torch.zeros(2 * len(ks) + 1, dtype=torch.int64, device=feats_a_local.device)        # hit counters + the non-finite word
    hits, flag = acc[: 2 * len(ks)].view(2, len(ks)), acc[2 * len(ks):]
    hip_sweep = topk is None
    # the finite check: the recall-only HIP paths report NaN / inf rows inside their counters (VTC_RECALL_NONFINITE: no launch of its own);
    # every other path gets one flag launch over this rank's rows
    rank_hip = hip_sweep and rank_ops is None and precision == 3 and RANK_PATH and len(ks) <= 4 and feats_a_local.is_cuda and same_width and (
        (world == 1 and n_total >= RANK_MIN_ROWS and d % 64 == 0) or (world > 1 and shard_ops is None and rank_sharded(n_total, d, precision, world, len(ks))))
    if not rank_hip:
        _flag_nonfinite(flag, feats_a_local, feats_b_local)
    if topk is None:
        from . import ops

        def topk(g, q, depth_):
            return ops.l2_topk(g, q, depth_, precision=precision, return_dists=False, ws=ws)[0]

    both, path, exchange = None, None, None
    if (hip_sweep and world == 1 and precision == 3 and RANK_PATH and n_total >= RANK_MIN_ROWS and len(ks) <= 4
            and feats_a_local.shape == feats_b_local.shape and d % 64 == 0):
        _world1_rank_path(a_all, b_all, ks, hits, ws, clock)
        path = sweep_path(n_total, precision, world, depth, d)
    elif hip_sweep and world == 1 and n_total >= (BIDIR_MIN_ROWS_F32 if precision == 0 else BIDIR_MIN_ROWS):
        both = _world1_one_matrix(a_all, b_all, depth, precision, ws, clock)
    elif world > 1 and (rank_ops is not None or (hip_sweep and shard_ops is None and same_width
                                                 and rank_sharded(n_total, d, precision, world, len(ks)))):
        path = sweep_path(n_total, precision, world, depth, d) if rank_ops is None else "injected rank ops"
        _rank_sharded(a_all, b_all, feats_a_local, feats_b_local, n_total, lo, ks, hits, rank, world, rank_ops, ws, clock)
        exchange = _exchange_name()
    elif world > 1 and (shard_ops is not None or (hip_sweep and one_matrix_sharded(n_total, precision, world, depth))):
        both = _one_matrix_sharded(a_all, b_all, feats_a_local, feats_b_local, n_total, depth, rank, world, shard_ops, ws, clock)
        exchange = _exchange_name()
    else:
        both = _two_searches(a_all, b_all, feats_a_local, feats_b_local, depth, topk, clock)
    if both is not None:
        _hits_from_ids(both, ks, lo, hi, hits)
        clock.mark("hits")
    if world > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        clock.mark("allreduce")
    vals = acc.cpu().tolist()            # ONE D2H; plain Python integers from here (torch CPU ops cost microseconds each: the 10k sweep is 0.28 ms)
    clock.close(path or (sweep_path(n_total, precision, world, depth, d) if hip_sweep else "injected top-k"), exchange)
    nh = 2 * len(ks)
    marker = rank_hip and any(v >> 40 for v in vals[:nh])              # VTC_RECALL_NONFINITE, summed over the ranks
    if vals[-1] != 0 or marker:
        raise ValueError("sharded_recall: non-finite values in the embeddings of at least one rank -- the ranks of such rows are undefined "
                         "(vtc_amd.host.model.nonfinite_cause lists what this build knows can produce them)")
    h = [v & ((1 << 40) - 1) for v in vals[:nh]] if rank_hip else vals[:nh]
    nk = len(ks)
    return ({k: h[j] / n_total for j, k in enumerate(k_vals)}, {k: h[nk + j] / n_total for j, k in enumerate(k_vals)})
