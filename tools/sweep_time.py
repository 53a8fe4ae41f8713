"""N x N EXACT sweep + R@K as bench.py times it (one rank): median / min / p90 of per-repetition host time and HIP-event time.
usage: python tools/sweep_time.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vtc_amd import _lib as L, ops, dist as vdist
torch.set_grad_enabled(False)
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
for N in [int(a) for a in sys.argv[1:]] or [10000, 50000]:
    g2 = torch.Generator(device=dev).manual_seed(123)
    va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=dev), dim=-1)
    noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=dev), dim=-1)
    tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2, device=dev), dim=-1)
    del noise
    ws = ops.workspace(vdist.sweep_workspace_bytes(N, N, 512, L.SWEEP_EXACT, 1), dev)
    for _ in range(3):
        r = vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws)
    reps = 40 if N <= 20000 else 12
    host, evs = [], []
    for i in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        r = vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws)
        e1.record()
        torch.cuda.synchronize()
        host.append(1e3 * (time.perf_counter() - t0))
        evs.append(e0.elapsed_time(e1))
    ph = {}
    vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws, phases=ph)
    print(f"N={N}: host median {np.median(host):.3f} ms (min {min(host):.3f}, p90 {np.percentile(host, 90):.3f}); gpu events median {np.median(evs):.3f} ms; "
          f"phases {ph}; R@K {r}", flush=True)
    del va, tb, ws
