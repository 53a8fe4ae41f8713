import os, sys, torch
sys.path.insert(0, os.getcwd())
from vtc_amd import ops
N = int(sys.argv[1])
g = torch.Generator().manual_seed(123)
a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
for _ in range(3):
    ops.recall_bidir(a, b, [1, 5, 10])
    torch.cuda.synchronize()
