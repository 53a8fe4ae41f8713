"""Per-(epilogue, N, K) GEMM time of one video-tower forward (HIP events around every launch): which instantiation costs what.
usage: python tools/tower_modes.py [B]    (env knobs of gemm.hip apply, e.g. VTC_GEMM_RESID_SMALL_K=768)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as BN
from oracle import arch as A
from vtc_amd import towers
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
a = A.VIT_B32
sd = A.synth_visual(a, 65, nframes=8, prefix="v.")
g = torch.Generator().manual_seed(1)
for k in list(sd):
    if k.endswith("temporal_fc.weight"):
        sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
pv = towers.PackedVision({k: v.cuda() for k, v in sd.items()}, "v.", torch.bfloat16)
vid = torch.randn(B, 8, 3, 224, 224, device="cuda").bfloat16()
for _ in range(2):
    pv.forward(vid)
torch.cuda.synchronize()
recs = BN.prof_records(lambda: [pv.forward(vid) for _ in range(2)], torch.cuda.current_stream().cuda_stream)
groups = {}
for x in recs:
    key = (x["cls"], x["region"]) + (x["tag"] if x["cls"].startswith("gemm") else ())
    e = groups.setdefault(key, [0.0, 0, 0.0])
    e[0] += x["ms"] / 2; e[1] += 1; e[2] += x["work"] / 2
tot = sum(v[0] for v in groups.values())
print(f"B={B}: kernel time per forward {tot:.2f} ms")
for key, (ms, n, work) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    name = key[0] + "/" + key[1] + (f" mode={BN.GEMM_MODE_NAMES.get(key[2], key[2])} N={key[3]} K={key[4]}" if len(key) > 2 else "")
    rate = f"{work / (ms * 1e-3) / 1e12:7.1f} TFLOP/s" if key[0].startswith("gemm") else f"{work / (ms * 1e-3) / 1e9:7.0f} GB/s"
    print(f"  {ms:8.3f} ms  {n // 2:3d} launches  avg {1e3 * ms / (n / 2):8.1f} us  {rate}  {name}")
