"""Does a process that ran the one-launch CAM tear down cleanly under rocprofv3?  usage: rocprofv3 --kernel-trace -- python3 tools/exit_probe.py fused|nofused"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.filterwarnings("ignore")
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
m = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
m.compute_dtype = torch.bfloat16
if sys.argv[1] == "nofused":
    m._pack()["cam"].w.flags = 1
g = torch.Generator().manual_seed(1)
img = torch.randn(4, 3, 224, 224, generator=g).to(dev).bfloat16()
out = m(img, synth_tokens(4, 77, g).to(dev), synth_tokens(20, 77, g).reshape(4, 5, 77).to(dev))
torch.cuda.synchronize()
print("done", sys.argv[1], float(out[2].sum()), flush=True)
