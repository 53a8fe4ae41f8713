"""Where does an occasional ~30 ms stall in the timed sweep repetitions come from (BENCH_r02: sweep_10000_ms 12.18 vs 0.63)?
Replays bench.py's order of events -- a config-3 forward (so the caching allocator holds GiB), del + empty_cache(), then the
10k x 10k sweep -- and prints every repetition's host time and GPU time (HIP events); under
    rocprofv3 --kernel-trace --hip-trace --output-format csv -d gpurun_out/sweep_trace -- python3 tools/sweep_stall_trace.py
the API trace shows what the host was inside during a slow repetition.  usage: python tools/sweep_stall_trace.py [N] [reps] [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vtc_amd import _lib as L, ops, dist as vdist
from vtc_amd.host import model as HM
torch.set_grad_enabled(False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
if B > 0:
    import warnings
    warnings.filterwarnings("ignore")
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().to(dev)
    vid = torch.randn(B, 8, 3, 224, 224, device=dev).bfloat16()
    g = torch.Generator().manual_seed(1)
    t = torch.randint(1, 49405, (B, 77), generator=g); t[:, 0] = 49406; t[:, 20] = 49407
    c = t.repeat(5, 1).reshape(B, 5, 77)
    for _ in range(3):
        out = m(vid, t.to(dev), c.to(dev))
    torch.cuda.synchronize()
    del m, vid, out
    torch.cuda.empty_cache()
for prec, name in ((L.SWEEP_EXACT, "exact"), (L.SWEEP_F32, "f32")):
    g2 = torch.Generator().manual_seed(123)
    va = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
    noise = torch.nn.functional.normalize(torch.randn(N, 512, generator=g2), dim=-1)
    tb = torch.nn.functional.normalize(va + 4.0 * noise * torch.rand(N, 1, generator=g2), dim=-1)
    va, tb = va.to(dev), tb.to(dev)
    ws = ops.workspace(vdist.sweep_workspace_bytes(N, N, 512, prec, 1), dev)
    for _ in range(3):
        vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=prec, ws=ws)
    torch.cuda.synchronize()
    host, gpu, evs = [], [], []
    for i in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=prec, ws=ws)
        e1.record()
        torch.cuda.synchronize()
        host.append(1e3 * (time.perf_counter() - t0))
        evs.append((e0, e1))
    gpu = [a.elapsed_time(b) for a, b in evs]
    med = float(np.median(host))
    print(f"{name} N={N}: host median {med:.3f} min {min(host):.3f} max {max(host):.3f} | gpu median {np.median(gpu):.3f} max {max(gpu):.3f}", flush=True)
    for i, (h, gg) in enumerate(zip(host, gpu)):
        if h > 2 * med:
            print(f"   outlier rep {i}: host {h:.3f} ms gpu {gg:.3f} ms", flush=True)
    del ws
    torch.cuda.empty_cache()
