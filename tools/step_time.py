"""Config-3 forward time at a few batch sizes (env knobs are read by the library once per process: run once per setting).
usage: python tools/step_time.py [B ...]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.filterwarnings("ignore")
from vtc_amd.host import model as HM
from vtc_amd.host.datasets import synth_tokens
torch.set_grad_enabled(False)
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
for blk in m.model.visual.transformer.resblocks:
    torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
m.compute_dtype = HM.default_compute_dtype()         # VTC_COMPUTE_DTYPE=bf16 (default) | f16 | f32: tools/ab_env.sh B "VTC_COMPUTE_DTYPE=bf16" "VTC_COMPUTE_DTYPE=f16"
PIX = m.compute_dtype
g = torch.Generator().manual_seed(123)
for B in [int(x) for x in sys.argv[1:]] or [1, 50, 1024]:
    gg = torch.Generator(device=dev).manual_seed(B)
    vid = torch.randn(B, 8, 3, 224, 224, generator=gg, device=dev).to(PIX)
    title = synth_tokens(B, 77, g).to(dev)
    comments = synth_tokens(B * 5, 77, g, empty_frac=0.1).reshape(B, 5, 77).to(dev)
    for _ in range(3):
        m(vid, title, comments)
    torch.cuda.synchronize()
    reps = 100 if B <= 8 else (30 if B <= 64 else 8)
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            m(vid, title, comments)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    print(f"B={B:5d}: {1e3 * best:8.3f} ms per forward, {B / best:8.1f} pairs/s  (compute dtype {str(PIX).split('.')[-1]}, VTC_GEMM_DEEP={os.environ.get('VTC_GEMM_DEEP', 'default')})", flush=True)
    del vid
