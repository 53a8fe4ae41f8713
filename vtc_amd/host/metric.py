"""Drop-in for the reference's ``model.metric.RecallAtK`` (model/metric.py:103-187).

``compute`` keeps the reference's contract -- numpy fp32 ``[N, D]`` in, ``[(k, recall)]`` out,
search depth ``max(k)+1``, hit test ``target in rp[:k]``, denominator ``len(features_a)`` -- but
the exact-L2 search runs in libvtc_hip.so (vtc_l2_topk) instead of faiss.GpuIndexFlatL2.
"""
from __future__ import annotations

import time
from collections.abc import Iterable

import numpy as np
import torch

from .. import _lib as L
from .. import ops

__all__ = ["RecallAtK", "BaseMetric"]


class BaseMetric:
    def __init__(self, name):
        self.name = name
        self.writer = None
        self.is_train = True
        self.is_val = True

    def set_writer(self, writer):
        self.writer = writer


class RecallAtK(BaseMetric):
    #: SWEEP_EXACT (default): split-bf16 candidate lists re-ranked with fp64 distances -- the neighbour ids of exact
    #: arithmetic on every row (faiss's fp32 search, metric.py:140-146, can only differ from them inside fp32
    #: near-ties), at 0.6x the time of SWEEP_F32 (fp32 MFMA, what faiss's useFloat16=False does);
    #: SWEEP_BF16X3 (split bf16, ~5e-7) and SWEEP_BF16 are the approximate modes
    precision = L.SWEEP_EXACT

    def __init__(self, name_a, name_b, k_vals=5, device=None):
        super().__init__("recall@k")
        if not isinstance(k_vals, Iterable):     # the reference's collections.Iterable (metric.py:106) is gone in py>=3.10
            k_vals = [k_vals]
        self.k_vals = list(k_vals)
        self.name_a, self.name_b = name_a, name_b
        self.is_train = False
        self.device = device
        self.reset()

    def reset(self):
        self._ws = None
        self.insert_index = 0
        self.features_a_list, self.features_b_list = [], []

    def update(self, loss, output, meta):
        """metric.py:123-135; embeddings stay on the GPU (no per-batch D2H)."""
        fa, fb = output[0].detach(), output[1].detach()
        if self.device is None:
            self.device = fa.device
        self.features_a_list.append(fa)
        self.features_b_list.append(fb)
        self.insert_index += fa.shape[0]

    def _dev(self):
        """The GPU the search runs on: the features' own device when they are on one, else the current one (the reference's
        callers hand CPU arrays to compute(), metric.py:126-127,177-180; the search itself has no CPU path)."""
        if self.device is not None and torch.device(self.device).type == "cuda":
            return torch.device(self.device)
        return torch.device("cuda", torch.cuda.current_device())

    def _nonfinite_error(self, bits=3):
        which = " and ".join(n for n, m in ((self.name_a, 1), (self.name_b, 2)) if bits & m)
        return ValueError(f"RecallAtK: non-finite values in the {which} features -- a NaN distance compares below nothing, the ranks "
                          "of such rows are undefined (faiss would return arbitrary ids for them); fix the embeddings "
                          "(vtc_amd.host.model.nonfinite_cause lists what this build knows can produce them)")

    def _prep(self, features_a, features_b, check=True):
        """fp32 [N, D] on the GPU; D zero-padded to the sweep's granule of 64 (squared L2 distances are unchanged;
        faiss.IndexFlatL2 takes any D), depth = max(k) + 1 (metric.py:145) capped by the gallery size."""
        a = torch.as_tensor(features_a, dtype=torch.float32).to(self._dev())
        b = torch.as_tensor(features_b, dtype=torch.float32).to(self._dev())
        if a.dim() != 2 or b.dim() != 2:
            raise ValueError("RecallAtK.compute expects 2-D [N, D] features (one caption per video, SURVEY 3.3)")
        if a.shape[1] != b.shape[1]:
            raise ValueError(f"RecallAtK: feature dims differ ({a.shape[1]} vs {b.shape[1]})")
        pad = -a.shape[1] % 64
        if pad:
            a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
        if self.check_finite and check:
            bits = ops.nonfinite_bits(a, b)
            if bits:
                raise self._nonfinite_error(bits)
        depth = min(int(np.max(self.k_vals) + 1), a.shape[0])
        if depth > 64:
            raise ValueError(f"RecallAtK: max(k_vals) + 1 = {depth} exceeds the sweep's list depth of 64 (one entry per lane of "
                             "a wavefront); the reference's callers use k in {1, 5, 10}")
        return a, b, depth

    def topk_ids(self, features_a, features_b) -> torch.Tensor:
        a, b, depth = self._prep(features_a, features_b)
        ids, _ = ops.l2_topk(a, b, depth, precision=self.precision, return_dists=False, ws=self._workspace(
            L.lib().vtc_l2_topk_workspace_bytes(a.shape[0], b.shape[0], a.shape[1], self.precision, 0), a.device))
        return ids

    def _workspace(self, nbytes, device):
        """One sweep workspace per metric object, grown on demand and dropped by reset() (a fresh multi-GiB
        allocation per search can land on a hipMalloc)."""
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < nbytes or ws.device != device:
            self._ws = None
            ws = self._ws = ops.workspace(nbytes, device)
        return ws

    def compute(self, features_a, features_b):
        """metric.py:137-161."""
        num_samples = features_a.shape[0]
        return self._hits_to_recall(self.topk_ids(features_a, features_b), num_samples)

    #: galleries at least this large take both directions from ONE distance matrix (vtc_l2_topk_bidir: the second
    #: direction is read off the columns of the blocks the first direction wrote).  Measured one-matrix vs two searches,
    #: EXACT: 10k 1.54 vs 1.33 ms, 16k 2.46 vs 2.65, 25k 5.1 vs 6.0, 50k 15.4 vs 21.1 (the column pass is issue-bound
    #: and pays a list initialisation per segment; the GEMM it saves grows with N^2)
    bidir_min_rows = 5120          # tools/bidir_threshold.py: one matrix wins from ~5k rows (EXACT 10k: 0.64 vs 0.79 ms)
    bidir_min_rows_f32 = 3000      # SWEEP_F32: the fp32-MFMA GEMM it saves is the expensive part at every size
    #: EXACT mode, paired rows: hit counters straight from the distance GEMM's key planes (the rank of each query's own gallery row),
    #: no sorted neighbour lists (round 5; the library takes n >= 1024)
    #: (VTC_SWEEP_RANK=0, read at import as vtc_amd.dist.RANK_PATH is: sorted neighbour lists instead -- the A/B knob of INTEGRATION.md)
    rank_path = __import__("os").environ.get("VTC_SWEEP_RANK", "1") != "0"
    rank_min_rows = 1024
    #: inputs are checked for NaN / inf before the search (one tiny launch + a 4-byte D2H) and rejected: a non-finite row would be
    #: ranked arbitrarily
    check_finite = True

    def _hits_to_recall(self, ids, num_samples):
        ks = [min(int(k), ids.shape[1]) for k in self.k_vals]
        out = []
        for i in range(0, len(ks), 4):
            hits = ops.recall_hits(ids, ks[i:i + 4]).cpu().numpy()
            out += [(k, float(h) / num_samples) for k, h in zip(self.k_vals[i:i + 4], hits)]
        return out

    def compute_both(self, features_a, features_b):
        """(compute(a, b), compute(b, a)) -- the two calls every caller of the reference makes back to back
        (metric.py:177-180, evaluation/eval.py:117-127, retrieval_evaluation.py:38-44)."""
        min_rows = self.bidir_min_rows_f32 if self.precision == L.SWEEP_F32 else self.bidir_min_rows
        if features_a.shape[0] != features_b.shape[0] or features_a.shape[0] < min(min_rows, self.rank_min_rows):
            return self.compute(features_a, features_b), self.compute(features_b, features_a)
        a, b, depth = self._prep(features_a, features_b, check=False)
        ks = [int(k) for k in self.k_vals]
        if (self.precision == L.SWEEP_EXACT and self.rank_path and a.shape[0] >= self.rank_min_rows and len(ks) <= 4
                and max(ks) <= a.shape[0] and ops.recall_bidir_supported(a.shape[0], a.shape[1])):
            # paired rows, parity mode: the reference asks only whether the query's own index is among the first k -- the RANK of one
            # gallery row -- so the sorted lists are never built (vtc_l2_recall_bidir; the same counters as the two-step form below).
            # The finite check rides in the counters (VTC_RECALL_NONFINITE): no launch, no D2H of its own.
            hits, bad = ops.split_recall_counters(ops.recall_bidir(a, b, ks, ws=self._workspace(
                L.lib().vtc_l2_recall_bidir_workspace_bytes(a.shape[0], a.shape[1]), a.device)).cpu())
            if bad and self.check_finite:
                raise self._nonfinite_error(ops.nonfinite_bits(a, b) or 3)
            hits = hits.numpy()
            n = a.shape[0]
            return ([(k, float(h) / n) for k, h in zip(self.k_vals, hits[0])], [(k, float(h) / n) for k, h in zip(self.k_vals, hits[1])])
        if features_a.shape[0] < min_rows:
            return self.compute(features_a, features_b), self.compute(features_b, features_a)
        if self.check_finite:
            bits = ops.nonfinite_bits(a, b)
            if bits:
                raise self._nonfinite_error(bits)
        ids_b2a, _, ids_a2b, _ = ops.l2_topk_bidir(a, b, depth, precision=self.precision, return_dists=False, ws=self._workspace(
            L.lib().vtc_l2_topk_bidir_workspace_bytes(a.shape[0], b.shape[0], a.shape[1], self.precision, 0), a.device))
        return self._hits_to_recall(ids_b2a, a.shape[0]), self._hits_to_recall(ids_a2b, b.shape[0])

    def avg(self):
        return None

    def result(self):
        """metric.py:166-187."""
        tic = time.time()
        fa, fb = torch.cat(self.features_a_list), torch.cat(self.features_b_list)
        assert self.insert_index == len(fa)
        res = {}
        r_ab, r_ba = self.compute_both(fa, fb)
        for k, r in r_ab:
            res[f"{self.name_b}_from_{self.name_a}-recall_at_{k}"] = r
        for k, r in r_ba:
            res[f"{self.name_a}_from_{self.name_b}-recall_at_{k}"] = r
        if self.writer:
            for name, r in res.items():
                self.writer.add_scalar(name, r)
        print("RecallAtK: result() took %.3fs" % (time.time() - tic))
        return res
