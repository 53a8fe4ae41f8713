"""Per-rank kernel phases of the sharded sweep at G ranks, measured on ONE card (the ranks' kernels run alone, one rank at a time; the
collectives are not measured here): for rank 0 of G, ms of (rows: prologue + [N/G, N] distance GEMM + finish of the row direction) and
(cols: finish of the column direction from G sources' planes), for the sorted-list form (vtc_l2_sweep_shard_*) and the recall-only form
(vtc_l2_recall_shard_*).  Inputs of DESIGN.md section 6's phase model.   usage: python tools/shard_phase_time.py [N ...] [--world G]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int, default=[10000, 50000])
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    from vtc_amd import dist as vdist
    from vtc_amd import ops
    G, ks, depth = args.world, [1, 5, 10], 11
    for n in args.sizes:
        rng = np.random.default_rng(n)
        a = rng.standard_normal((n, 512)).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = a + 0.02 * rng.standard_normal((n, 512)).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        bounds = [vdist.shard_bounds(n, r, G) for r in range(G)]
        rb = ops.sweep_row_block()
        nbp = -(-max(h - l for l, h in bounds) // rb)
        lo, hi = bounds[0]
        bl, al = tb[lo:hi].contiguous(), ta[lo:hi].contiguous()
        ws = ops.workspace(vdist.sweep_workspace_bytes(n, hi - lo, 512, 3, G), ta.device)
        hits = torch.zeros(2, 3, dtype=torch.int64, device="cuda")
        planes = [ops.recall_shard_rows(ta, tb[l:h].contiguous(), l, ks, nbp, hits[0], ws=ws) for l, h in bounds]
        recv = torch.stack([pl[:, :, lo:hi] for pl in planes]).contiguous()
        planes4 = [ops.sweep_shard_rows(ta, tb[l:h].contiguous(), depth, nbp, ws=ws)[1] for l, h in bounds]      # the sorted-list form's four planes
        recv4 = torch.stack([pl[:, :, lo:hi] for pl in planes4]).contiguous()
        sb = torch.tensor([l for l, _ in bounds], dtype=torch.int32, device="cuda")
        sbn = torch.tensor([l for l, _ in bounds] + [n], dtype=torch.int32, device="cuda")
        out = {
            "rows_gemm_select (sorted lists)": timed(lambda: ops.sweep_shard_rows(ta, bl, depth, nbp, ws=ws)),
            "cols_select (sorted lists)": timed(lambda: ops.sweep_shard_cols(tb, al, depth, recv4, sb, ws=ws)),
            "rows_gemm_rank (recall only)": timed(lambda: ops.recall_shard_rows(ta, bl, lo, ks, nbp, hits[0], ws=ws)),
            "cols_rank (recall only)": timed(lambda: ops.recall_shard_cols(tb, al, lo, ks, recv, sbn, hits[1], ws=ws)),
        }
        print(f"N={n} G={G} (rank 0: {hi - lo} rows; planes to exchange {planes[0].shape[0] * nbp * n * 4 / 1e6:.2f} MB per rank (recall only; sorted lists: {4 * nbp * n * 4 / 1e6:.2f})): "
              + "; ".join(f"{k} {v:.3f} ms" for k, v in out.items()), flush=True)


if __name__ == "__main__":
    main()
