"""Config 3 (8-frame TimeSformer + title + 5 comments + CAM, B = 256) forward passes for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
torch.manual_seed(1023)
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
for blk in m.model.visual.transformer.resblocks:
    torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
m = m.eval().to(dev)
m.compute_dtype = torch.bfloat16
m.overlap_towers = os.environ.get("VTC_OVERLAP", "1") != "0"
g = torch.Generator().manual_seed(123)
B = int(os.environ.get("B", "256"))
vid = torch.randn(B, 8, 3, 224, 224, generator=g).to(dev).bfloat16()
title = synth_tokens(B, 77, g).to(dev)
comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
for _ in range(4):
    m(vid, title, comments)
torch.cuda.synchronize()
