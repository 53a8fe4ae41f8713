"""16-frame TimeSformer (BASELINE stress config) at a serving batch: finite outputs, batch independence, pairs/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd.host import model as HM
torch.set_grad_enabled(False)
torch.manual_seed(1023)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
HM.PretrainedCLIP_TimeSformer_finaltf.nframes = 16
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().cuda()
g = torch.Generator().manual_seed(123)
vis = torch.randn(B, 16, 3, 224, 224, generator=g).bfloat16().cuda()
L = torch.randint(1, 76, (B * 6,), generator=g)
tok = torch.zeros(B * 6, 77, dtype=torch.long)
for i, l in enumerate(L.tolist()):
    tok[i, 0] = 49406; tok[i, 1:1 + l] = torch.randint(1, 49406, (l,), generator=g); tok[i, 1 + l] = 49407
title, comments = tok[:B].cuda(), tok[B:].reshape(B, 5, 77).cuda()
out = m(vis, title, comments)
assert all(torch.isfinite(o).all() for o in out)
sub = m(vis[:8], title[:8], comments[:8])
print("batch independence (max abs diff, vis/text):", float((out[0][:8] - sub[0]).abs().max()), float((out[1][:8] - sub[1]).abs().max()))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): m(vis, title, comments)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"F=16 B={B}: {dt*1e3:.1f} ms/step, {B/dt:.0f} pairs/s")
