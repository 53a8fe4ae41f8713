"""RCCL smoke on ONE rank (what a 1-GPU box can run of the multi-GPU path): process-group init with backend "nccl", all_reduce,
all_gather_into_tensor and the all_to_all over unbound views that vtc_amd/dist.py:exchange_column_planes issues.
usage: python tools/rccl_single_rank.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
try:
    from vtc_amd import dist as D
    x = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(x)
    assert x.tolist() == list(range(8))
    rows = torch.randn(5, 16, device="cuda")
    out = rows.new_empty(5, 16)
    dist.all_gather_into_tensor(out, rows)
    assert torch.equal(out, rows)
    planes = torch.randint(0, 1 << 30, (4, 3, 40), dtype=torch.int32, device="cuda")
    got = D.exchange_column_planes(planes, 40, 0, 1)
    assert got.shape == (1, 4, 3, 40) and torch.equal(got[0], planes)
    torch.cuda.synchronize()
    print("rccl single-rank OK: backend", dist.get_backend(), "all_reduce / all_gather_into_tensor / all_to_all(unbind views)")
finally:
    dist.destroy_process_group()
