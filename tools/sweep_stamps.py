"""50k sweep GEMMs under the -DVTC_GEMM_STAMPS build (VTC_HIP_LIB=vtc_amd/lib/variants/libvtc_stamps.so): per-tile phase cycles on stderr."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vtc_amd import _lib as L, ops
N = 50000
g = torch.Generator().manual_seed(123)
a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
for prec in (L.SWEEP_BF16, L.SWEEP_BF16X3):
    for _ in range(2):
        ops.l2_topk(a, b, 11, precision=prec, return_dists=False)
        torch.cuda.synchronize()
