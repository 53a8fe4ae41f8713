"""Two GEMM launches for PMC collection: 118272x1536 with K=512 (short) and K=4096 (steady state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
for K in (512, 4096):
    a = (torch.randn(118272, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(1536, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(1536, device="cuda")
    out = torch.zeros(118272, 1536, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, b, epilogue=L.EPI_STORE, out=out)
    torch.cuda.synchronize()
