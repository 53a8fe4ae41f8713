import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vtc_amd import _lib as L, ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(1)
a = rng.standard_normal((n, 512)).astype(np.float32); a /= np.linalg.norm(a, axis=1, keepdims=True)
ta = torch.from_numpy(a).cuda()
ids, d = ops.l2_topk(ta, ta, 11, precision=L.SWEEP_EXACT)
bad = (ids[:, 0].cpu() != torch.arange(n)).nonzero().flatten()
print("n", n, "rows not finding themselves:", bad.numel(), bad[:20].tolist())
for r in bad[:3].tolist():
    print(r, ids[r].tolist(), d[r].tolist())
