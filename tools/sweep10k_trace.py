"""10k x 10k EXACT sweep + R@K, as bench.py runs it: wall time per call and (under rocprofv3 --kernel-trace) the launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops, dist as vdist
torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g2 = torch.Generator().manual_seed(123)
va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2), dim=-1)
va, tb = va.cuda(), tb.cuda()
ws = ops.workspace(vdist.sweep_workspace_bytes(N, N, 512, L.SWEEP_EXACT, 1), va.device)
for _ in range(3):
    vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws)
torch.cuda.synchronize()
for reps in (1, 10):
    t0 = time.perf_counter()
    for _ in range(reps):
        r = vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws)
    torch.cuda.synchronize()
    print(f"N={N}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call ({reps} calls)", flush=True)
# GPU-side only: events around the C call
ids1 = torch.empty(N, 11, dtype=torch.int64, device="cuda"); ids2 = torch.empty_like(ids1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.l2_topk_bidir(va, tb, 11, precision=L.SWEEP_EXACT, return_dists=False, ws=ws)
e1.record(); torch.cuda.synchronize()
print(f"vtc_l2_topk_bidir alone (events, 10 calls): {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
