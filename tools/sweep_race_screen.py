"""Race screen for the sweep's distance GEMM with the two-plane epilogue (EPI_L2MIN2: half norms through LDS, written behind one tile's epilogue and
read before the next tile's K loop) and the rank kernels behind it: the RAW key planes of repeated launches on the same inputs must be bit-identical
(every key is a pure function of the inputs: any stale read of the norm area or of an LDS stage shows as a differing word), and so must the
counters, under a streaming side load that moves the timing.  Sizes: ragged edge tiles, one / several tiles per workgroup.
usage: python tools/sweep_race_screen.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vtc_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
hog_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB
hog_b = torch.empty_like(hog_a)
bad_total = 0
for (n, d, world) in [(4099, 512, 3), (10000, 512, 8), (20011, 256, 4), (50000, 512, 8)]:
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, d)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = (a + (0.9 / np.sqrt(d)) * rng.standard_normal((n, d))).astype(np.float32)
    ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    ks = [1, 5, 10]
    ref = ops.recall_bidir(ta, tb, ks).clone()
    # one rank's share of the sharded form: its planes are returned to the caller, so they can be compared word for word
    lo, hi = 0, -(-n // world)
    rb = ops.sweep_row_block()
    nbp = -(-(hi - lo) // rb)
    h0 = torch.zeros(3, dtype=torch.int64, device=dev)
    planes_ref = ops.recall_shard_rows(ta, tb[lo:hi].contiguous(), lo, ks, nbp, h0).clone()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    r_here = reps if n <= 20011 else max(20, reps // 5)
    for r in range(r_here):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a, non_blocking=True)
        got = ops.recall_bidir(ta, tb, ks)
        bad += (got != ref).any().long()
        h = torch.zeros(3, dtype=torch.int64, device=dev)
        pl = ops.recall_shard_rows(ta, tb[lo:hi].contiguous(), lo, ks, nbp, h)
        bad += (pl != planes_ref).any().long() + (h != h0).any().long()
    torch.cuda.synchronize()
    k = int(bad.item())
    bad_total += k
    print(f"n={n:6d} d={d:4d}: {r_here} x (vtc_l2_recall_bidir + one rank's vtc_l2_recall_shard_rows of {world}), planes {tuple(planes_ref.shape)}: "
          f"differing launches {k}; hits {ref.tolist()}", flush=True)
print("total differing launches:", bad_total)
sys.exit(1 if bad_total else 0)
