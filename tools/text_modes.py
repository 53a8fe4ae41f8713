"""Per-kernel time of the text tower alone (HIP events around every launch): device-side ragged bookkeeping (default), host offsets,
dense; IEEE-half and bf16 blocks.  usage: python tools/text_modes.py [n_sequences]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.filterwarnings("ignore")
import bench as BN
from vtc_amd import towers
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.set_num_threads(8)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
m = HM.PretrainedCLIP_TimeSformer(model_type="ViT-B/32")
sd = {"t." + k[len("model."):]: v.detach().cuda() for k, v in m.state_dict().items()
      if k.startswith("model.") and not k.startswith("model.visual.")}
g = torch.Generator().manual_seed(124)
ids = synth_tokens(S, 77, g, empty_frac=0.08).cuda()
print(f"{S} sequences, {int((ids.argmax(-1) + 1).sum())} rows of {S * 77}")
stream = torch.cuda.current_stream().cuda_stream


def show(name, fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    R = 3
    recs = BN.prof_records(lambda: [fn() for _ in range(R)], stream)
    groups = {}
    for x in recs:
        key = (x["cls"], x["region"]) + (x["tag"] if x["cls"].startswith("gemm") else ())
        e = groups.setdefault(key, [0.0, 0, 0.0])
        e[0] += x["ms"] / R; e[1] += 1; e[2] += x["work"] / R
    tot = sum(v[0] for v in groups.values())
    print(f"== {name}: kernel time per forward {tot:.3f} ms, {len(recs) // R} launches")
    for key, (ms, n, work) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:8]:
        nm = key[0] + "/" + key[1] + (f" mode={BN.GEMM_MODE_NAMES.get(key[2], key[2])} N={key[3]} K={key[4]}" if len(key) > 2 else "")
        rate = f"{work / (ms * 1e-3) / 1e12:7.1f} TFLOP/s" if key[0].startswith("gemm") else f"{work / (ms * 1e-3) / 1e9:7.0f} GB/s"
        print(f"  {ms:8.3f} ms  {n // R:3d} launches  avg {1e3 * ms / (n / R):8.1f} us  {rate}  {nm}")


for half in (12, 0):
    pk = towers.PackedText(sd, "t.", torch.bfloat16, heads=8, half_layers=half)
    show(f"half_layers={half}, ragged, device-side bookkeeping", lambda: pk.forward(ids))
    show(f"half_layers={half}, ragged, host offsets", lambda: pk.forward_host_offsets(ids))
    if half == 12:
        show(f"half_layers={half}, dense", lambda: pk.forward(ids, ragged=False))
