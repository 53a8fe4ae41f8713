"""EXACT sweep: block-minima path (v2) vs split-bf16 materialising path (v1), per tile variant; ids vs fp64 oracle sample.
usage: python tests/probes/sweep_v2_bench.py N    (env VTC_SWEEP_EXACT_V1=1 / VTC_SWEEP_MIN_TILE=0|1 / VTC_SWEEP_DEBUG=1)"""
import os, sys, time
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
def bench(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
t2, (i1, _, i2, _) = bench(lambda: ops.l2_topk_bidir(a, b, 11, precision=L.SWEEP_EXACT, return_dists=False))
t1, (j1, _) = bench(lambda: ops.l2_topk(a, b, 11, precision=L.SWEEP_EXACT, return_dists=False))
rows = np.arange(3, N, max(1, N // 500))[:500]
r1 = E.l2_topk(va.numpy(), tb.numpy()[rows], 11, np.float64)[0]
r2 = E.l2_topk(tb.numpy(), va.numpy()[rows], 11, np.float64)[0]
ok = bool(np.array_equal(i1.cpu().numpy()[rows], r1) and np.array_equal(i2.cpu().numpy()[rows], r2) and np.array_equal(j1.cpu().numpy()[rows], r1))
print(f"N={N} EXACT: bidir {t2:.3f} ms | one direction {t1:.3f} ms | sample == fp64 oracle: {ok} | v1={os.environ.get('VTC_SWEEP_EXACT_V1','0')} tile={os.environ.get('VTC_SWEEP_MIN_TILE','0')}", flush=True)
