"""Three-shape GEMM timing for kernel-variant comparisons (see tools/build_variant.sh).
usage: VTC_HIP_LIB=... python tools/gemm_quick.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtc_amd import _lib as L
from vtc_amd import ops

lib = L.lib()
stream = torch.cuda.current_stream().cuda_stream
SHAPES = [(8192, 8192, 8192, L.EPI_STORE, "8192^3"), (402432, 2304, 768, L.EPI_STORE, "video qkv"), (402432, 3072, 768, L.EPI_GELU, "video c_fc"), (402432, 768, 768, L.EPI_RESID, "video out"), (402432, 768, 3072, L.EPI_RESID, "video c_proj"), (118272, 512, 512, L.EPI_RESID, "out K=512"), (118272, 1536, 512, L.EPI_STORE, "qkv K=512"),
          (118272, 2048, 512, L.EPI_GELU, "c_fc K=512"), (118272, 512, 2048, L.EPI_RESID, "c_proj K=2048")]
res = []
for M, N, K, epi, label in SHAPES:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID else torch.bfloat16)
    for _ in range(2):
        ops.gemm(a, w, b, epilogue=epi, out=out)
    torch.cuda.synchronize()
    lib.vtc_prof_begin()
    for _ in range(5):
        ops.gemm(a, w, b, epilogue=epi, out=out)
    n = len(L.PROF_CLASSES)
    ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    lib.vtc_prof_end(stream, ms, cnt, work)
    t = ms[0] / 5
    res.append(f"{label}: {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:6.0f} TF")
    del a, w, out
print(" | ".join(res), flush=True)
