"""Debug probe: out = A @ I^T must reproduce A; prints where it does not (row / 8-column chunk pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vtc_amd import ops
for dt in (torch.float32, torch.bfloat16):
    for (M, N, K) in ((128, 128, 128), (512, 512, 256), (2048, 4096, 256), (1000, 700, 192), (8192, 4096, 512)):
        a = torch.arange(M * K, dtype=torch.float32).reshape(M, K) % 251
        w = torch.zeros(N, K); w[torch.arange(min(N, K)), torch.arange(min(N, K))] = 1
        out = ops.gemm(a.cuda().to(dt), w.cuda().to(dt), None, out_dtype=torch.float32).cpu()
        ref = a @ w.t()
        bad = (out != ref)
        print(dt, M, N, K, "bad elements:", int(bad.sum()), "of", bad.numel())
        if bad.any():
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("  bad rows (first 24):", rows[:24].tolist(), " n_bad_rows", len(rows))
            print("  bad cols (first 24):", cols[:24].tolist(), " n_bad_cols", len(cols))
            r = int(rows[0]); print("  row", r, "got", out[r, :16].tolist(), "\n        exp", ref[r, :16].tolist())
