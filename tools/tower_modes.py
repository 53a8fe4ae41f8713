"""Per-(epilogue, N, K) GEMM time of one video-tower forward (HIP events around every launch): which instantiation costs what.
usage: python tools/tower_modes.py [B]    (env knobs of gemm.hip apply, e.g. VTC_GEMM_RESID_SMALL_K=768)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as BN
import warnings
warnings.filterwarnings("ignore")
from vtc_amd import towers
from vtc_amd.host import model as HM
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(65)
m = HM.PretrainedCLIP_TimeSformer(model_type="ViT-B/32")
for blk in m.model.visual.transformer.resblocks:       # trained weights are not the init's zeros
    torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
sd = {"v." + k[len("model.visual."):]: v.detach() for k, v in m.state_dict().items() if k.startswith("model.visual.")}
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
