"""Per-(class, region, epilogue, N, K) kernel time of one config-3 forward at batch B (HIP events around every launch, towers serialised).
usage: python tools/step_modes.py [B]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.filterwarnings("ignore")
import bench as BN
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.set_num_threads(8)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
m.compute_dtype = torch.bfloat16
m.overlap_towers = False
g = torch.Generator().manual_seed(123)
vid = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev).bfloat16()
title = synth_tokens(B, 77, g).to(dev)
comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
for _ in range(3):
    m(vid, title, comments)
torch.cuda.synchronize()
R = 4
recs = BN.prof_records(lambda: [m(vid, title, comments) for _ in range(R)], torch.cuda.current_stream().cuda_stream)
groups = {}
for x in recs:
    key = (x["cls"], x["region"]) + (x["tag"] if x["cls"].startswith("gemm") else ())
    e = groups.setdefault(key, [0.0, 0, 0.0])
    e[0] += x["ms"] / R; e[1] += 1; e[2] += x["work"] / R
tot = sum(v[0] for v in groups.values())
print(f"B={B}: kernel time per forward {tot:.3f} ms, {len(recs) // R} launches")
for key, (ms, n, work) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
    name = key[0] + "/" + key[1] + (f" mode={BN.GEMM_MODE_NAMES.get(key[2], key[2])} N={key[3]} K={key[4]}" if len(key) > 2 else "")
    rate = f"{work / (ms * 1e-3) / 1e12:7.1f} TFLOP/s" if key[0].startswith("gemm") else f"{work / (ms * 1e-3) / 1e9:7.0f} GB/s"
    print(f"  {ms:8.3f} ms  {n // R:3d} launches  avg {1e3 * ms / (n / R):8.1f} us  {rate}  {name}")
