"""Context Adapter Module alone (vtc_cam_forward): time per call and kernel launches per call at the reference's operating points.
usage: python tools/cam_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import arch as A
from vtc_amd import _lib as L, towers
torch.set_grad_enabled(False)
a = A.VIT_B32
sd = A.synth_cam(a, 3)
pk = towers.PackedCam({k: v.cuda() for k, v in sd.items()}, torch.float32, 8, True, None)
lib = L.lib()
for B in (1, 8, 50, 128, 256, 1024):
    g = torch.Generator().manual_seed(B)
    main = torch.randn(B, 512, generator=g).cuda()
    comm = torch.randn(B * 5, 512, generator=g).cuda()
    comments = A.synth_tokens(B * 5, a, 5, empty_frac=0.2).reshape(B, 5, -1).cuda()
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
