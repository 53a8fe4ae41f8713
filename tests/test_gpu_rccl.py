"""What ONE GPU can run of the multi-GPU path's RCCL calls (backend "nccl" of torch.distributed is RCCL on ROCm): process-group
init, all_reduce, all_gather_into_tensor, and the all_to_all over unbound views that vtc_amd/dist.py issues for the sharded
sweep's column planes -- with world_size 1.  The bookkeeping of world > 1 (shard bounds, per-source row bases, padded blocks) is
covered on CPU under gloo (tests/test_dist_gloo.py); an 8-GPU run is the driver's."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_single_rank_collectives_of_the_sharded_sweep():
    import torch.distributed as dist
    from vtc_amd import dist as D
    assert not dist.is_initialized()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        x = torch.arange(8, dtype=torch.float32, device="cuda")
        dist.all_reduce(x)
        assert x.tolist() == list(range(8))
        rows = torch.randn(5, 16, device="cuda")
        out = rows.new_empty(5, 16)
        dist.all_gather_into_tensor(out, rows)
        assert torch.equal(out, rows)
        planes = torch.randint(0, 1 << 30, (4, 3, 40), dtype=torch.int32, device="cuda")
        got = D.exchange_column_planes(planes, 40, 0, 1)          # dist.all_to_all(list(recv.unbind(0)), send)
        assert got.shape == (1, 4, 3, 40) and torch.equal(got[0], planes)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
