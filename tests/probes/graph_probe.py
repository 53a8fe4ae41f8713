"""Small-batch towers: per-launch enqueue vs one hipGraph replay of the same C-ABI call (torch.cuda.CUDAGraph capturing
vtc_vision_forward / vtc_text_forward_ragged).  usage: python tests/probes/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vtc_amd import towers
from oracle import arch as A
torch.set_grad_enabled(False)
a = A.VIT_B32
pv = towers.PackedVision({k: v.cuda() for k, v in A.synth_visual(a, 65, nframes=8, prefix="v.").items()}, "v.", torch.bfloat16)
pt = towers.PackedText({k: v.cuda() for k, v in A.synth_text(a, 62, prefix="t.").items()}, "t.", torch.bfloat16, heads=a.transformer_heads)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for B in (1, 4, 8, 32, 64):
    vid = torch.randn(B, 8, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
    txt = A.synth_tokens(6 * B, a, 67, empty_frac=0.1).cuda()
    ref_v, ref_t = pv.forward(vid), pt.forward(txt)
    t_v, t_t = timeit(lambda: pv.forward(vid)), timeit(lambda: pt.forward(txt))
    torch.cuda.synchronize()
    gv, gt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gv):
        out_v = pv.forward(vid)
    with torch.cuda.graph(gt):
        out_t = pt.forward(txt, ragged=False)
    gv.replay(); gt.replay(); torch.cuda.synchronize()
    ok = torch.equal(out_v, ref_v) and torch.equal(out_t, ref_t)
    g_v, g_t = timeit(gv.replay), timeit(gt.replay)
    print(f"B={B:3d}: video tower {t_v:.3f} ms -> graph {g_v:.3f} ms | text tower ({6 * B} seq) {t_t:.3f} -> {g_t:.3f} ms | identical: {ok}", flush=True)
