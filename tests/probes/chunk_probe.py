"""Vision tower throughput vs VISION_CHUNK (activations of a chunk resident in the 256 MiB Infinity Cache?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vtc_amd import towers
from oracle import arch as A
torch.set_grad_enabled(False)
a = A.VIT_B32
sd = {k: v.cuda() for k, v in A.synth_visual(a, 65, nframes=8, prefix="v.").items()}
pv = towers.PackedVision(sd, "v.", torch.bfloat16)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
vid = torch.randn(B, 8, 3, 224, 224, device="cuda").bfloat16()
for chunk in (0, 256, 128, 64, 48, 32, 16, 0):
    towers.VISION_CHUNK = chunk
    pv.forward(vid); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2): pv.forward(vid)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f"B={B} chunk={chunk or B}: {dt*1e3:.1f} ms  {B/dt:.0f} videos/s", flush=True)
