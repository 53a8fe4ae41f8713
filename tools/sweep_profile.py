"""N x N EXACT sweep + R@K as bench.py runs it (vdist.sharded_recall on one rank), for
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/sweep_profile.py N reps
usage: python tools/sweep_profile.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops, dist as vdist
torch.set_grad_enabled(False)
torch.set_num_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
g2 = torch.Generator(device=dev).manual_seed(123)
va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=dev), dim=-1)
noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2, device=dev), dim=-1)
tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2, device=dev), dim=-1)
del noise
ws = ops.workspace(vdist.sweep_workspace_bytes(N, N, 512, L.SWEEP_EXACT, 1), dev)
for _ in range(3 + reps):
    r = vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=L.SWEEP_EXACT, ws=ws)
torch.cuda.synchronize()
print(f"N={N}: {3 + reps} sweeps, R@1/5/10 {r}")
