"""Folded LayerNorm under mean-dominated rows: a DC offset on ln_pre.bias puts |mean| / std of every residual row at the given
ratio.  Error vs the fp32 oracle with the fold on and off.  usage: python tests/probes/fold_dc_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vtc_amd import _lib as L, towers
from oracle import arch as A, clip_ref as CR
torch.set_grad_enabled(False)
a = A.VIT_B32
lib = L.lib()
def unit(x): return x / np.linalg.norm(x, axis=-1, keepdims=True)
img = A.synth_pixels((4, 3, 224, 224), 7)
for dc in (0.0, 1.0, 3.0, 10.0):
    sd = A.synth_visual(a, 104, prefix="v.")
    sd["v.ln_pre.bias"] = sd["v.ln_pre.bias"] + dc
    ref = unit(CR.encode_image(img, sd, a, "v.").numpy())
    pk = towers.PackedVision({k: v.cuda() for k, v in sd.items()}, "v.", torch.bfloat16)
    out = []
    for on in (0, 1):
        pk.w.flags = towers.tower_flags(ln_fold=bool(on))
        got = unit(pk.forward(img.cuda()).cpu().numpy())
        out.append(np.abs(got - ref).max())
    print(f"dc offset {dc:5.1f}: max err vs fp32 oracle  LayerNorm kernels {out[0]:.2e} | folded {out[1]:.2e}", flush=True)
