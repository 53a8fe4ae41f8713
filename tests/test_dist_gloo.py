"""CPU, world_size 2, gloo: the sharded-sweep bookkeeping of vtc_amd/dist.py (shard bounds, ragged
all-gather, target offsets, counter all-reduce) against the oracle's unsharded Recall@K.
The search itself is injected (oracle's CPU search); on the GPU box the HIP sweep takes its place."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import eval_ref as E


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpu_topk(g, q, depth):
    ids, _ = E.l2_topk(g.numpy(), q.numpy(), depth)
    return torch.from_numpy(ids)


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from vtc_amd import dist as vdist
    r, _, w = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(0)
    a = rng.standard_normal((n, 32)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + 0.7 * rng.standard_normal((n, 32)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    lo, hi = vdist.shard_bounds(n, rank, world)
    r_ab, r_ba = vdist.sharded_recall(torch.from_numpy(a[lo:hi]), torch.from_numpy(b[lo:hi]), n, [1, 5, 10], rank, world,
                                      topk=_cpu_topk)
    if rank == 0:
        ref_ab = dict(E.recall_at_k(a, b, [1, 5, 10]))
        ref_ba = dict(E.recall_at_k(b, a, [1, 5, 10]))
        out.put((r_ab == ref_ab, r_ba == ref_ba, r_ab, ref_ab))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 101])          # even and ragged shards
def test_sharded_recall_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_ab, ok_ba, got, ref = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_ab and ok_ba, (got, ref)


def test_shard_bounds_cover_everything():
    from vtc_amd.dist import shard_bounds
    for n in (1, 7, 8, 10000, 50001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
