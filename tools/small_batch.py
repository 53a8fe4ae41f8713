"""Config-3 forward at small batches: ms per forward and launches (events + host clock).  usage: python tools/small_batch.py [B ...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.filterwarnings("ignore")
from vtc_amd import _lib as L
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
m.compute_dtype = torch.bfloat16
g = torch.Generator().manual_seed(123)
for B in [int(x) for x in sys.argv[1:]] or [1, 8, 50]:
    vid = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev).bfloat16()
    title = synth_tokens(B, 77, g).to(dev)
    comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
    for _ in range(5):
        m(vid, title, comments)
    torch.cuda.synchronize()
    n0 = L.lib().vtc_debug_launch_count()
    m(vid, title, comments)
    nl = L.lib().vtc_debug_launch_count() - n0
    reps = 100
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m(vid, title, comments)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"B={B:4d}: {1e3 * dt:7.3f} ms per forward, {B / dt:8.1f} pairs/s, {nl} launches", flush=True)
