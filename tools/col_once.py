"""One vtc_l2_topk_bidir call (BF16) at N = argv[1] (default 10000): the workload for rocprofv3 --pmc passes over col_topk_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator().manual_seed(123)
a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
ops.l2_topk_bidir(a, b, 11, precision=L.SWEEP_BF16, return_dists=False)
torch.cuda.synchronize()
