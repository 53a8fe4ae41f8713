import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vtc_amd import _lib as L, ops
from oracle import eval_ref as E
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator().manual_seed(123)
va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1)
noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1)
tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g), dim=-1)
a, b = va.cuda(), tb.cuda()
i1, d1, i2, d2 = ops.l2_topk_bidir(a, b, 11, precision=L.SWEEP_EXACT)
j1, e1 = ops.l2_topk(a, b, 11, precision=L.SWEEP_EXACT)
j2, e2 = ops.l2_topk(b, a, 11, precision=L.SWEEP_EXACT)
# reference: F32 mode (materialising path)
k1, _ = ops.l2_topk(a, b, 11, precision=L.SWEEP_BF16X3)
k2, _ = ops.l2_topk(b, a, 11, precision=L.SWEEP_BF16X3)
for name, x, y in (("bidir rows vs x3", i1, k1), ("bidir cols vs x3", i2, k2), ("single a,b vs x3", j1, k1), ("single b,a vs x3", j2, k2), ("bidir rows vs single", i1, j1), ("bidir cols vs single", i2, j2)):
    bad = (x != y).any(dim=1).nonzero().flatten()
    print(name, "mismatching rows:", bad.numel(), bad[:10].tolist())
    if bad.numel():
        r = int(bad[0])
        print("   row", r, "got", x[r].tolist(), "\n   ref", y[r].tolist())
