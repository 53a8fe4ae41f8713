"""A/B of VTC_TOWER_FUSED_TIME on the video tower alone (config 3's tower, B videos): per-forward time and the attention-region
kernel time with every row of the last block computed (what bench.py's timesformer_attention_mfma_frac prices).
usage: python tools/fused_time_ab.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as BN
from oracle import arch as A
from vtc_amd import _lib as L, towers
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sd = {k: v.cuda() for k, v in A.synth_visual(A.VIT_B32, 5, nframes=8, prefix="v.").items()}
for k in list(sd):
    if k.endswith("temporal_fc.weight"):
        sd[k] = torch.randn(sd[k].shape, device="cuda") * 0.02
vid = BN.gpu_randn((B, 8, 3, 224, 224), 123, torch.device("cuda"), torch.bfloat16)
stream_ptr = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for ft in (False, True):
        pk = towers.PackedVision(sd, "v.", torch.bfloat16)
        pk.w.flags = towers.tower_flags(full_last_layer=True, fused_time=ft)
        for _ in range(2):
            pk.forward(vid)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            pk.forward(vid)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5 * 1e3
        pv = BN.prof_regions(lambda: pk.forward(vid), stream_ptr)
        attn_ms = sum(v["attn"]["ms"] for v in pv.values())
        frac = 4.27e9 * 12 * B / (attn_ms * 1e-3) / 2.5e15
        print(f"fused_time={int(ft)}: tower forward {dt:.2f} ms; attention region {attn_ms:.2f} ms -> mfma frac {frac:.4f}; "
              + ", ".join(f"{k} {sum(v[r]['ms'] for r in v):.2f}" for k, v in pv.items() if sum(v[r]['launches'] for r in v)), flush=True)
