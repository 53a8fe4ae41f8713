"""Yardstick helper: the rocBLAS / hipBLASLt kernels PyTorch-ROCm picks for the tower GEMM shapes (run under
`rocprofv3 --kernel-trace --stats` to get their names = macro tile / depth / wave layout, and durations).  Tools only."""
import sys

import torch

shapes = [(8192, 8192, 8192), (402432, 2304, 768), (402432, 3072, 768), (402432, 768, 768), (402432, 768, 3072)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.mm(a, w.t(), out=out)
    torch.cuda.synchronize()
    print(M, N, K, "done", flush=True)
