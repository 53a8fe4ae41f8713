"""Sweep micro-benchmark: vtc_l2_topk on N x N unit-norm embeddings, per-kernel-class time from the
library's HIP-event facility.  usage: python tools/sweep_bench.py [N ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vtc_amd import _lib as L
from vtc_amd import ops

lib = L.lib()
stream = torch.cuda.current_stream().cuda_stream
for N in [int(a) for a in sys.argv[1:]] or [10000, 50000]:
    g = torch.Generator().manual_seed(123)
    a = torch.nn.functional.normalize(torch.randn(N, 512, generator=g), dim=-1).cuda()
    b = torch.nn.functional.normalize(a.cpu() + 0.5 * torch.randn(N, 512, generator=g) / 22.6, dim=-1).cuda()
    for name, prec in (("f32", L.SWEEP_F32), ("bf16x3", L.SWEEP_BF16X3), ("bf16", L.SWEEP_BF16), ("exact", L.SWEEP_EXACT)):
        for rpb in [int(x) for x in os.environ.get('RPB', '0').split(',')]:
            ops.l2_topk(a, b, 11, precision=prec, rows_per_block=rpb, return_dists=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ops.l2_topk(a, b, 11, precision=prec, rows_per_block=rpb, return_dists=False)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            lib.vtc_prof_begin()
            ops.l2_topk(a, b, 11, precision=prec, rows_per_block=rpb, return_dists=False)
            n = len(L.PROF_CLASSES)
            ms, cnt, work = (C.c_double * n)(), (C.c_longlong * n)(), (C.c_double * n)()
            lib.vtc_prof_end(stream, ms, cnt, work)
            d = {L.PROF_CLASSES[i]: (round(ms[i], 3), cnt[i]) for i in range(n) if cnt[i]}
            gemm_ms = ms[0] + ms[1]
            print(f"N={N} {name:7s} rpb={rpb} one direction: wall {wall*1e3:8.3f} ms | {d} | gemm {2.0*N*N*512/gemm_ms/1e9:7.1f} TFLOP/s (x3 K for bf16x3) | "
                  f"topk {4.0*N*N/ms[5]/1e6:7.1f} GB/s | matrix traffic 8N^2/wall = {8.0*N*N/wall/1e9:7.1f} GB/s", flush=True)
