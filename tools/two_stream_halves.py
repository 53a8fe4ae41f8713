"""two half-batches of the video tower on two streams, so that one half's HBM-bound launches (attention cores,
K = N = 768 residual GEMMs) can run under the other half's MFMA-bound GEMMs.  Round 2 measured this on that round's kernels
(profiles/r02_experiments.txt: 113.2 -> 113.4 - 117.5 ms); this re-measures it on the current ones.
  VTC_GEMM_CU_BUDGET=n (read once per process) caps every persistent 256 x 256 GEMM grid at n workgroups = CUs: with n = 128 two GEMMs are
  co-resident, with 160 / 192 a GEMM leaves 96 / 64 CUs to the other stream's memory-bound kernels.
usage: [VTC_GEMM_CU_BUDGET=n] python tools/two_stream_halves.py [B] [stagger_rows...]
Prints ms per B-video tower forward: one stream; two streams x B/2 started together; second stream started `stagger` items late."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd.host import model as HM
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
torch.manual_seed(1023)
m = HM.PretrainedCLIP_TimeSformer(model_type="ViT-B/32").eval().to(dev)
for blk in m.model.visual.transformer.resblocks:
    torch.nn.init.normal_(blk.temporal_fc.weight, std=0.02)
m.compute_dtype = torch.bfloat16
pv = m._pack()["visual"]
g = torch.Generator(device=dev).manual_seed(5)
vid = torch.randn(B, 8, 3, 224, 224, generator=g, device=dev).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    return pv.forward(vid)


def two(split):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = pv.forward(vid[:split])
    with torch.cuda.stream(s2):
        b = pv.forward(vid[split:])
    cur.wait_stream(s1); cur.wait_stream(s2)
    return torch.cat([a, b])


def timed(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps, out


budget = os.environ.get("VTC_GEMM_CU_BUDGET", "0")
t1, ref = timed(one)
print(f"CU budget {budget}: one stream, B={B}: {t1:.2f} ms")
for split in [B // 2] + [int(x) for x in sys.argv[2:]]:
    t2, out = timed(lambda: two(split))
    print(f"CU budget {budget}: two streams [{split}, {B - split}]: {t2:.2f} ms   max |diff| vs one stream {float((out - ref).abs().max()):.1e}")
