"""GEMM micro-benchmark: TFLOP/s of vtc_gemm per shape/epilogue, timed with the library's own
HIP-event facility.  usage: python tools/gemm_bench.py [bf16|f32]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtc_amd import _lib as L
from vtc_amd import ops

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
lib = L.lib()
stream = torch.cuda.current_stream().cuda_stream
SHAPES = [
    # (M, N, K, epilogue, label)
    (4096, 4096, 4096, L.EPI_STORE, "square 4096^3 store"),
    (8192, 8192, 8192, L.EPI_STORE, "square 8192^3 store"),
    (4096, 4096, 16384, L.EPI_STORE, "deep K=16384 store"),
    (118272, 1536, 512, L.EPI_STORE, "text qkv"),
    (118272, 512, 512, L.EPI_RESID, "text out_proj resid"),
    (118272, 2048, 512, L.EPI_GELU, "text c_fc gelu"),
    (118272, 512, 2048, L.EPI_RESID, "text c_proj resid"),
    (12800, 2304, 768, L.EPI_STORE, "vit qkv"),
    (12800, 768, 3072, L.EPI_RESID, "vit c_proj resid"),
    (100608, 2304, 768, L.EPI_STORE, "tsf qkv B=256"),
    (100608, 768, 768, L.EPI_RESID, "tsf out_proj resid"),
    (100608, 3072, 768, L.EPI_GELU, "tsf c_fc gelu"),
    (100608, 768, 3072, L.EPI_RESID, "tsf c_proj resid"),
]
for M, N, K, epi, label in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID else dt)
    for _ in range(2):
        ops.gemm(a, w, b, epilogue=epi, out=out)
    torch.cuda.synchronize()
    lib.vtc_prof_begin()
    reps = 5
    for _ in range(reps):
        ops.gemm(a, w, b, epilogue=epi, out=out)
    n = len(L.PROF_CLASSES)
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    lib.vtc_prof_end(stream, ms, cnt, work)
    i = 0 if dt == torch.bfloat16 else 1
    t = ms[i] / reps
    print(f"{label:28s} M={M:6d} N={N:5d} K={K:5d}  {t*1e3:9.1f} us  {2.0*M*N*K/t/1e9:8.1f} TFLOP/s", flush=True)
    del a, w, out
