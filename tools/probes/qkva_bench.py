"""Fused QKV+attention kernel vs QKV GEMM + attention kernel, per attention-branch shape (one process, interleaved).
usage: python tools/qkva_bench.py [B]      (the VTC_QKVA_SKIP phase ablations of round 2 are gone from the kernel: round 3 removed the wrong-answer knobs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import ops
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(0)

def bench(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for name, heads, n_seq, L, kw, rows in (
        ("time  F=8 ", 12, B * 49, 8, dict(s2=49, a0=1, a1=393, a2=8, a3=0, pstride=1), B * 393),
        ("space F=8 ", 12, B * 8, 50, dict(s2=8, a0=0, a1=393, a2=0, a3=1, pstride=8), B * 393),
        ("ViT   L=50", 12, B * 8, 50, dict(), B * 8 * 50),
        ("text  L=77", 8, B * 6, 77, dict(causal=True), B * 6 * 77)):
    W = heads * 64
    h = torch.randn(rows, W, generator=g).cuda().bfloat16()
    wq = (torch.randn(3 * W, W, generator=g) * W ** -0.5).cuda().bfloat16()
    bq = torch.randn(3 * W, generator=g).cuda()
    qkv = torch.empty(rows, 3 * W, dtype=torch.bfloat16, device="cuda")
    o1, o2 = torch.zeros(rows, W, dtype=torch.bfloat16, device="cuda"), torch.zeros(rows, W, dtype=torch.bfloat16, device="cuda")
    cls = torch.zeros(n_seq, W, device="cuda") if name.startswith("space") else None
    t_g = bench(lambda: ops.gemm(h, wq, bq, out=qkv))
    t_a = bench(lambda: ops.attention(qkv, n_seq, L, heads, cls_out=cls, out=o1, **kw))
    t_f = bench(lambda: ops.qkv_attention(h, wq, bq, n_seq, L, heads, cls_out=cls, out=o2, **kw))
    fl = 2.0 * rows * 3 * W * W
    print(f"{name} rows {rows:7d} W {W}: gemm {t_g:.3f} ms ({fl/t_g/1e9:.0f} TF) + attn {t_a:.3f} ms = {t_g+t_a:.3f} | fused {t_f:.3f} ms ({fl/t_f/1e9:.0f} TF)", flush=True)
