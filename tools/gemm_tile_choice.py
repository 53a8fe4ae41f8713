"""Calibrate the tile-configuration heuristic: time vision-tower shapes under VTC_GEMM_TILE (run once per value)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
lib = L.lib(); stream = torch.cuda.current_stream().cuda_stream
SH = [(12800, 768, 768, L.EPI_RESID), (12800, 768, 3072, L.EPI_RESID), (12800, 3072, 768, L.EPI_GELU), (12800, 2304, 768, L.EPI_STORE),
      (25152, 768, 768, L.EPI_RESID), (25152, 768, 3072, L.EPI_RESID), (25152, 3072, 768, L.EPI_GELU), (25152, 2304, 768, L.EPI_STORE),
      (6400, 768, 3072, L.EPI_RESID), (3200, 768, 3072, L.EPI_RESID), (4096, 4096, 4096, L.EPI_STORE), (1536, 512, 512, L.EPI_STORE)]
out_s = []
for M, N, K, epi in SH:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16(); b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID else torch.bfloat16)
    for _ in range(2): ops.gemm(a, w, b, epilogue=epi, out=out)
    torch.cuda.synchronize(); lib.vtc_prof_begin()
    for _ in range(5): ops.gemm(a, w, b, epilogue=epi, out=out)
    n = len(L.PROF_CLASSES); ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    lib.vtc_prof_end(stream, ms, cnt, work)
    out_s.append(f"{M}x{N}x{K}:{ms[0]/5*1e3:6.1f}")
print(" ".join(out_s), flush=True)
