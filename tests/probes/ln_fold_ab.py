"""Folded LayerNorm A/B (vtc_*_w.flags, VTC_TOWER_NO_LN_FOLD): tower outputs against each other and step time.  usage: python tests/probes/ln_fold_ab.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vtc_amd import _lib as L, towers
from oracle import arch as A
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
a = A.VIT_B32
lib = L.lib()
def unit(x): return x / x.norm(dim=-1, keepdim=True)
def cuda_sd(sd): return {k: v.cuda() for k, v in sd.items()}
sdv = A.synth_visual(a, 65, nframes=8, prefix="v.")
pv = towers.PackedVision(cuda_sd(sdv), "v.", torch.bfloat16)
pv32 = towers.PackedVision(cuda_sd(sdv), "v.", torch.float32)
sdt = A.synth_text(a, 62, prefix="t.")
pt = towers.PackedText(cuda_sd(sdt), "t.", torch.bfloat16, heads=a.transformer_heads)
pt32 = towers.PackedText(cuda_sd(sdt), "t.", torch.float32, heads=a.transformer_heads)
vid = A.synth_pixels((min(B, 64), 8, 3, 224, 224), 66).cuda().bfloat16()
txt = A.synth_tokens(6 * min(B, 256), a, 64, empty_frac=0.1).cuda()
ref_v, ref_t = unit(pv32.forward(vid[:8].float())), unit(pt32.forward(txt[:96]))
for on in (0, 1):
    pv.w.flags = pt.w.flags = towers.tower_flags(ln_fold=bool(on))
    v, t = unit(pv.forward(vid)), unit(pt.forward(txt))
    print(f"fold={on}: video vs fp32 max {(v[:8] - ref_v).abs().max().item():.2e} rms {(v[:8] - ref_v).pow(2).mean().sqrt().item():.2e} | "
          f"text vs fp32 max {(t[:96] - ref_t).abs().max().item():.2e} rms {(t[:96] - ref_t).pow(2).mean().sqrt().item():.2e}", flush=True)
vb = torch.randn(B, 8, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
tb = A.synth_tokens(6 * B, a, 67, empty_frac=0.1).cuda()
for rep in range(2):
    for on in (0, 1):
        pv.w.flags = pt.w.flags = towers.tower_flags(ln_fold=bool(on))
        pv.forward(vb); pt.forward(tb); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): pv.forward(vb)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(3): pt.forward(tb)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"fold={on}: video tower {1e3 * (t1 - t0) / 3:.2f} ms, text tower {1e3 * (t2 - t1) / 3:.2f} ms (B={B})", flush=True)
