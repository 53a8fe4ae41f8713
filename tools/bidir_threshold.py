"""EXACT sweep, both directions: two vtc_l2_topk searches vs one vtc_l2_topk_bidir, by N (sets BIDIR_MIN_ROWS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops
torch.set_grad_enabled(False)
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for N in (1024, 2048, 3000, 4096, 6000, 8192, 10000, 14336, 20000):
    g = torch.Generator().manual_seed(N)
    a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
    b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1), dim=-1).cuda()
    ws = ops.workspace(L.lib().vtc_l2_topk_bidir_workspace_bytes(N, N, 512, L.SWEEP_EXACT, 0), a.device)
    for prec, name in ((L.SWEEP_EXACT, "exact"), (L.SWEEP_F32, "f32")):
        two = t(lambda: (ops.l2_topk(a, b, 11, precision=prec, return_dists=False, ws=ws), ops.l2_topk(b, a, 11, precision=prec, return_dists=False, ws=ws)))
        one = t(lambda: ops.l2_topk_bidir(a, b, 11, precision=prec, return_dists=False, ws=ws))
        print(f"N={N:6d} {name:5s}: two searches {two:.3f} ms | bidir {one:.3f} ms", flush=True)
