"""Phase-level cycle stamps of the deep K loop (diagnostic build: tools/build_variant.sh pstamps -DVTC_GEMM_PHASE_STAMPS):
    VTC_HIP_LIB=vtc_amd/lib/variants/libvtc_pstamps.so python3 tools/phase_stamps.py
prints, per launch, the cycles a wave spends inside an MFMA cluster and between two clusters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
for (M, N, K) in ((8192, 8192, 8192), (402432, 2304, 768), (402432, 768, 3072)):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(6):
        ops.gemm(a, w, b, epilogue=L.EPI_STORE, out=out)
    torch.cuda.synchronize()
    del a, w, out
