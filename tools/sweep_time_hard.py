"""The recall-only sweep on data where the targets are NOT at the top (planted pairs under heavy noise: R@1 well below 1, so many queries have
a handful of closer entries and the rank kernels' fp64 side has work): ms per sweep with two and with four planes.  usage: python tools/sweep_time_hard.py [N] [noise ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vtc_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
noises = [float(x) for x in sys.argv[2:]] or [0.6, 3.0, 9.0, 30.0]
for noise in noises:
    rng = np.random.default_rng(7)
    a = rng.standard_normal((n, 512)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    z = rng.standard_normal((n, 512)).astype(np.float32)
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    b = a + noise * z * rng.uniform(0.2, 1.8, (n, 1)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    hits = torch.zeros(2, 3, dtype=torch.int64, device="cuda")
    ws = ops.workspace(1 << 20, ta.device)
    for _ in range(3):
        ops.recall_bidir(ta, tb, [1, 5, 10], hits=hits)
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        hits.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.recall_bidir(ta, tb, [1, 5, 10], hits=hits); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"N={n} noise={noise}: {np.median(ts):.3f} ms per sweep (planes: {os.environ.get('VTC_SWEEP_PLANES', 'default')}); R@1/5/10 = {(hits[0].cpu().numpy() / n).round(4).tolist()}", flush=True)
