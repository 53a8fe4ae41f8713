"""Three GEMM launches each of two shapes for PMC collection (rocprofv3 --pmc ... -- python3 tools/gemm_pmc.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
for (M, N, K) in ((8192, 8192, 8192), (118272, 1536, 512)):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, b, epilogue=L.EPI_STORE, out=out)
    torch.cuda.synchronize()
