"""Per-call wall time of the EXACT sweep at N = 10000 (bench.py's planted data), 30 calls: looks for intermittent slow calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops
N = 10000
g2 = torch.Generator().manual_seed(123)
va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2), dim=-1)
a, b = va.cuda(), tb.cuda()
for prec in (L.SWEEP_EXACT, L.SWEEP_BF16X3):
    ts = []
    for i in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.l2_topk(a, b, 11, precision=prec, return_dists=False)
        ops.l2_topk(b, a, 11, precision=prec, return_dists=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(prec, " ".join(f"{t:.2f}" for t in ts), flush=True)
