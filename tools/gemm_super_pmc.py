"""Three launches of the folded-QKV-shaped GEMM (M = 402 432, N = 2304, K = 768, bf16 store) for PMC collection:
    VTC_GEMM_SUPER=<row tiles per super-row> rocprofv3 --pmc FETCH_SIZE -- python3 tools/gemm_super_pmc.py
VERDICT r3 next #3: fetched bytes against time as the tile walk's super-row changes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vtc_amd import _lib as L
from vtc_amd import ops
M, N, K = 402432, 2304, 768
a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
b = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    ops.gemm(a, w, b, epilogue=L.EPI_STORE, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.gemm(a, w, b, epilogue=L.EPI_STORE, out=out)
e1.record()
torch.cuda.synchronize()
print(f"SUPER={os.environ.get('VTC_GEMM_SUPER', 'default(4)')} CG={os.environ.get('VTC_GEMM_CG', 'default(all)')}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch", flush=True)
