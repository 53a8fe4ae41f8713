"""Steady-state config-3 forwards for rocprofv3 --kernel-trace: tools/summarize_step_trace.py then checks that no kernel
foreign to libvtc_hip.so (an at::native::* torch kernel) runs between the first and the last launch of a forward
(VERDICT r2 #7).  usage: python tools/step_kernels.py [B] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
import torch
warnings.filterwarnings("ignore")
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
m.compute_dtype = torch.bfloat16
g = torch.Generator().manual_seed(123)
vid = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev).bfloat16()
title = synth_tokens(B, 77, g).to(dev)
comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
for _ in range(3):
    m(vid, title, comments)
torch.cuda.synchronize()
for _ in range(steps):
    out = m(vid, title, comments)
torch.cuda.synchronize()
print("done", float(out[2].sum()))
