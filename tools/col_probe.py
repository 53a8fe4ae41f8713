"""One search (vtc_l2_topk) vs both directions from one matrix (vtc_l2_topk_bidir), BF16 and EXACT, at N = argv[1]: prints ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator().manual_seed(123)
a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
for prec in (L.SWEEP_BF16, L.SWEEP_EXACT):
    for fn in (lambda: ops.l2_topk(a, b, 11, precision=prec, return_dists=False), lambda: ops.l2_topk_bidir(a, b, 11, precision=prec, return_dists=False)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); print(prec, f"{(time.perf_counter()-t0)/5*1e3:.3f} ms", flush=True)
