import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
lib = L.lib(); stream = torch.cuda.current_stream().cuda_stream
out_s = []
for dt in (torch.float32, torch.bfloat16):
  for M, N, K, epi in [(1536, 512, 512, L.EPI_STORE), (1536, 1536, 512, L.EPI_STORE), (1536, 2048, 512, L.EPI_GELU), (1536, 512, 2048, L.EPI_RESID), (256, 512, 768, L.EPI_STORE), (256, 256, 512, L.EPI_STORE)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(dt); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); b = torch.randn(N, device="cuda")
    out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == L.EPI_RESID else dt)
    for _ in range(3): ops.gemm(a, w, b, epilogue=epi, out=out)
    torch.cuda.synchronize(); lib.vtc_prof_begin()
    for _ in range(10): ops.gemm(a, w, b, epilogue=epi, out=out)
    n = len(L.PROF_CLASSES); ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
    lib.vtc_prof_end(stream, ms, cnt, work)
    out_s.append(f"{str(dt)[6:10]} {M}x{N}x{K}:{(ms[0]+ms[1])/10*1e3:6.1f}")
print(" | ".join(out_s), flush=True)
