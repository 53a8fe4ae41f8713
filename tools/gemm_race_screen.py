"""Race screen for the 256 x 256 GEMM's K-loop schedules (guide 5: "a sync-structure edit makes a NEW template: screen it for races over
many runs at several sizes").  Integer-valued operands make every output exact in fp32, so ANY stale LDS read / early re-fill shows as
a mismatch; a side stream streams large copies meanwhile so that LDS-DMA return times vary.
usage: python tools/gemm_race_screen.py [reps]        (VTC_GEMM_DEEP=0|1 selects the loop; default = the product's)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
hog_a = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB
hog_b = torch.empty_like(hog_a)
SHAPES = [(8192, 3072, 512), (8200, 3000, 768), (16384, 1536, 1024), (12800, 2304, 128), (12800, 2304, 192), (100608, 2304, 768),
          (100608, 768, 3072), (65536, 512, 2048)]
bad_total = 0
for (M, N, K) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), generator=g, device=dev).float()
    w = torch.randint(-3, 4, (N, K), generator=g, device=dev).float() + (torch.arange(N, device=dev).float()[:, None] % 3)
    bias = torch.randint(-5, 6, (N,), generator=g, device=dev).float()
    ref = a @ w.t() + bias                               # |values| < 2^24: exact
    ad, wd = a.bfloat16(), w.bfloat16()
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    x0 = torch.randint(-8, 9, (M, N), generator=g, device=dev).float()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for r in range(reps):
        if r % 4 == 0:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a, non_blocking=True)
        ops.gemm(ad, wd, bias, out_dtype=torch.float32, out=out)
        bad += (out != ref).any().long()
        if r % 8 == 0:                                    # the residual epilogue (reads + rewrites the output)
            x = x0.clone()
            ops.gemm(ad, wd, bias, epilogue=L.EPI_RESID, out=x)
            bad += (x != x0 + ref).any().long()
    torch.cuda.synchronize()
    n = int(bad.item())
    bad_total += n
    print(f"M={M:6d} N={N:5d} K={K:5d}: {reps} launches (+{(reps + 7) // 8} residual), mismatching launches: {n}", flush=True)
    del a, w, ref, out, x0
print("VTC_GEMM_DEEP =", os.environ.get("VTC_GEMM_DEEP", "default"), "-> total mismatching launches:", bad_total)
sys.exit(1 if bad_total else 0)
