"""Context Adapter Module alone (vtc_cam_forward): time per call and kernel launches per call at the reference's operating points.
usage: python tools/cam_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import warnings
warnings.filterwarnings("ignore")
from vtc_amd import _lib as L, towers
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.manual_seed(3)
m = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
for blk in m.final_transformer.resblocks:       # trained-like (non-zero) projections
    torch.nn.init.normal_(blk.attn.out_proj.weight, std=0.02)
    torch.nn.init.normal_(blk.mlp.c_proj.weight, std=0.02)
sd = {k: v.detach() for k, v in m.state_dict().items() if k.startswith(("final_transformer.", "final_linear.", "mask_embedding"))}
pk = towers.PackedCam({k: v.cuda() for k, v in sd.items()}, torch.float32, 8, True, None)
lib = L.lib()
for B in (1, 8, 50, 128, 256, 1024):
    g = torch.Generator().manual_seed(B)
    main = torch.randn(B, 512, generator=g).cuda()
    comm = torch.randn(B * 5, 512, generator=g).cuda()
    comments = synth_tokens(B * 5, 77, g, empty_frac=0.2).reshape(B, 5, -1).cuda()
    for _ in range(5):
        out = pk.forward(main, comm, comments)
    torch.cuda.synchronize()
    n0 = lib.vtc_debug_launch_count()
    pk.forward(main, comm, comments)
    nl = lib.vtc_debug_launch_count() - n0
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        pk.forward(main, comm, comments)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        pk.forward(main, comm, comments)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B:5d}: {1e6 * dt:8.1f} us per call (host loop), {1e3 * e0.elapsed_time(e1) / reps:8.1f} us (events), {nl} launches", flush=True)
