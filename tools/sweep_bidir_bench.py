"""Both-direction sweep: two vtc_l2_topk calls vs one vtc_l2_topk_bidir.  usage: python tools/sweep_bidir_bench.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtc_amd import _lib as L
from vtc_amd import ops

for N in [int(x) for x in sys.argv[1:]] or [10000, 50000]:
    g = torch.Generator().manual_seed(123)
    a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
    b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
    for name, prec in (("exact", L.SWEEP_EXACT), ("f32", L.SWEEP_F32), ("bf16x3", L.SWEEP_BF16X3), ("bf16", L.SWEEP_BF16)):
        def two():
            return ops.l2_topk(a, b, 11, precision=prec, return_dists=False)[0], ops.l2_topk(b, a, 11, precision=prec, return_dists=False)[0]

        def one():
            r = ops.l2_topk_bidir(a, b, 11, precision=prec, return_dists=False)
            return r[0], r[2]
        res = {}
        for label, fn in (("two searches", two), ("bidir", one)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = fn()
            torch.cuda.synchronize()
            res[label] = ((time.perf_counter() - t0) / 3 * 1e3, out)
        same = [float((x == y).float().mean()) for x, y in zip(res["two searches"][1], res["bidir"][1])]
        print(f"N={N} {name:7s}: two searches {res['two searches'][0]:8.3f} ms | bidir {res['bidir'][0]:8.3f} ms | ids equal {same}", flush=True)
