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

# ---- round 2's protocol, many times, in three variants that differ ONLY in what happens to device memory between measurements:
#   "free-to-driver"  a 12 GB block is allocated, touched, freed and handed back to the driver (torch.cuda.empty_cache(): hipFree)
#                     -- what bench.py did between the headline model and the sweeps;
#   "cached"          the same allocation churn stays inside torch's caching allocator (no hipMalloc / hipFree);
#   "pageable-h2d"    no device-memory churn, but 2 x 20 MB of PAGEABLE host memory are drawn, copied to the device and dropped
#                     (what run_sweep did with its synthetic embeddings in round 2); "pinned-h2d": the same from pinned memory;
#   "cpu-ops"         no device memory traffic at all: the synthetic embeddings are drawn and normalised on the HOST with torch's
#                     default intra-op thread count (what run_sweep did before its timed repetitions); "cpu-ops-4t": 4 threads;
#   "none"            no churn.
# Beside each variant: the cgroup's CPU-bandwidth throttling counters (cpu.stat nr_throttled / throttled_usec) over it.
# Each measurement: ONE warm-up, three timed repetitions (round 2); a repetition > 3 x the fastest counts as a stall.
cycles = int(os.environ.get("VTC_STALL_CYCLES", "12"))
gb = int(os.environ.get("VTC_STALL_GB", "12"))
va = torch.nn.functional.normalize(torch.randn(N, 512, generator=torch.Generator().manual_seed(123)), dim=-1).to(dev)
tb = torch.nn.functional.normalize(torch.randn(N, 512, generator=torch.Generator().manual_seed(124)), dim=-1).to(dev)
def throttle():
    try:
        kv = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
    except OSError:
        return 0, 0


print(f"torch intra-op threads {torch.get_num_threads()}, os.cpu_count {os.cpu_count()}, cpu.max {open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else '?'}", flush=True)
nthreads0 = torch.get_num_threads()
for variant in ("none", "cached", "free-to-driver", "pageable-h2d", "pinned-h2d", "cpu-ops", "cpu-ops-4t", "none"):
    th0 = throttle()
    torch.set_num_threads(4 if variant == "cpu-ops-4t" else nthreads0)
    slow, worst = 0, 0.0
    torch.cuda.synchronize()
    for cyc in range(cycles):
        if variant.startswith("cpu-ops"):
            for seed in (1, 2, 3):
                h = torch.nn.functional.normalize(torch.randn(N, 512, generator=torch.Generator().manual_seed(seed)), dim=-1)
                del h
        elif variant.endswith("h2d"):
            for seed in (1, 2):
                h = torch.randn(N, 512, generator=torch.Generator().manual_seed(seed))
                if variant == "pinned-h2d":
                    h = h.pin_memory()
                d_ = h.to(dev, non_blocking=variant == "pinned-h2d")
                torch.cuda.synchronize()
                del h, d_
        elif variant != "none":
            big = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
            big.fill_(1)
            torch.cuda.synchronize()
            del big
            if variant == "free-to-driver":
                torch.cuda.empty_cache()
        for prec, name in ((L.SWEEP_EXACT, "exact"), (L.SWEEP_F32, "f32"), (L.SWEEP_BF16X3, "bf16x3")):
            ws = ops.workspace(vdist.sweep_workspace_bytes(N, N, 512, prec, 1), dev)
            vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=prec, ws=ws)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                vdist.sharded_recall(va, tb, N, [1, 5, 10], 0, 1, precision=prec, ws=ws)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            if max(ts) > 3 * min(ts):
                slow += 1
                worst = max(worst, max(ts))
            del ws
    th1 = throttle()
    print(f"variant {variant:15s}: {slow} stalled measurements in {3 * cycles} (worst repetition {worst:.1f} ms) | cgroup throttled "
          f"{th1[0] - th0[0]} periods, {(th1[1] - th0[1]) / 1e3:.1f} ms", flush=True)
