"""Residual GEMM + following LayerNorm: one launch (EPI_RESID_LN) vs two.  usage: python tools/resid_ln_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L, ops
torch.set_grad_enabled(False)
g = torch.Generator().manual_seed(0)
def bench(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, M, N, K, skip in (("vision out-proj B=1024", 402432, 768, 768, 393), ("vision c_proj B=1024", 402432, 768, 3072, 0),
                            ("vision out-proj B=256", 100608, 768, 768, 393), ("text out-proj 1536 seq", 118272, 512, 512, 0),
                            ("text c_proj 1536 seq", 118272, 512, 2048, 0)):
    a = torch.randn(M, K, generator=g).cuda().bfloat16()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().bfloat16()
    b = torch.randn(N, generator=g).cuda(); lg = torch.randn(N, generator=g).cuda(); lb = torch.randn(N, generator=g).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    h = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    t_g = bench(lambda: ops.gemm(a, w, b, epilogue=L.EPI_RESID, out=x, skip_mod=skip))
    t_l = bench(lambda: ops.layernorm(x, lg, lb, out_dtype=torch.bfloat16))
    def two():
        ops.gemm(a, w, b, epilogue=L.EPI_RESID, out=x, skip_mod=skip); ops.layernorm(x, lg, lb, out_dtype=torch.bfloat16)
    t_2 = bench(two)
    t_f = bench(lambda: ops.gemm_resid_layernorm(a, w, b, x, lg, lb, skip_mod=skip, ln_out=h))
    t_g2 = bench(lambda: ops.gemm(a, w, b, epilogue=L.EPI_RESID, out=x, skip_mod=skip))
    print(f"{name:24s}: gemm {t_g:.3f} + ln {t_l:.3f} (back to back {t_2:.3f}) ms | fused {t_f:.3f} ms | gemm again {t_g2:.3f}", flush=True)
