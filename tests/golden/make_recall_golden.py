"""Generate tests/golden/recall_cases.npz by running the REFERENCE's own ``RecallAtK`` (model/metric.py:103-187,
unmodified) on constructed rank cases.

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_recall_golden.py [/root/reference]

``RecallAtK`` cannot run as shipped here: ``faiss-gpu`` (environment.yml:12) is not installed and needs a GPU, and
``collections.Iterable`` (metric.py:106) is gone in Python >= 3.10.  Two stand-ins, for this script only:
  * ``collections.Iterable = collections.abc.Iterable``;
  * a numpy ``faiss`` module with the four names metric.py touches (GpuIndexFlatConfig, StandardGpuResources,
    GpuIndexFlatL2.add/.search).  Its search is faiss's published exact-L2 definition (squared L2, fp32, k smallest,
    ascending); ties -- unspecified in faiss -- by lowest index.
What this pins is everything the reference's OWN code decides: search depth ``max(k)+1`` (:145), the hit rule
``target in rp[:k]`` (:153-155), the denominator ``len(features_a)`` (:138,158), ``result()``'s key names and which
direction is which (:177-180), and the list bookkeeping of ``update`` (:123-135).  The stand-in's own arithmetic is
NOT evidence about faiss; the fixture stores inputs and the reference's outputs.
"""
from __future__ import annotations

import collections
import collections.abc
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

collections.Iterable = collections.abc.Iterable          # metric.py:106

faiss = types.ModuleType("faiss")


class GpuIndexFlatConfig:
    useFloat16 = False
    device = 0


class StandardGpuResources:
    pass


class GpuIndexFlatL2:
    def __init__(self, res, dim, cfg):
        assert cfg.useFloat16 is False                    # metric.py:113
        self.dim, self.x = dim, np.zeros((0, dim), dtype=np.float32)

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape[1] == self.dim
        self.x = np.concatenate([self.x, x])

    def search(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        d = (q * q).sum(1)[:, None] + (self.x * self.x).sum(1)[None, :] - np.float32(2) * (q @ self.x.T)
        order = np.argsort(d, axis=1, kind="stable")[:, :k]
        if order.shape[1] < k:                            # faiss pads missing neighbours with -1
            pad = -np.ones((q.shape[0], k - order.shape[1]), dtype=order.dtype)
            return np.take_along_axis(d, order, 1), np.concatenate([order, pad], 1)
        return np.take_along_axis(d, order, 1), order


faiss.GpuIndexFlatConfig, faiss.StandardGpuResources, faiss.GpuIndexFlatL2 = GpuIndexFlatConfig, StandardGpuResources, GpuIndexFlatL2
sys.modules["faiss"] = faiss
# the reference's file, loaded as-is by path (importing the ``model`` package would pull in model/model.py -> ``clip``)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_model_metric", os.path.join(REF, "model", "metric.py"))
ref_metric = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_metric)
assert ref_metric.__file__.startswith(REF), ref_metric.__file__


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def planted(n, d, ranks, seed):
    """query i sits at a chosen rank of its own gallery row: rank - 1 other rows are strictly closer"""
    rng = np.random.default_rng(seed)
    a = unit(rng.standard_normal((n, d))).astype(np.float32)
    b = np.empty_like(a)
    for i in range(n):
        q = a[i].copy()
        for j in [j for j in range(n) if j != i][: ranks[i % len(ranks)] - 1]:
            q = q + 1.5 * a[j]
        b[i] = q
    return a, b


cases = {}
rng = np.random.default_rng(0)
a, b = planted(64, 32, (1, 2, 5, 6, 10, 11), 0)
cases["planted_ranks"] = (a, b, [1, 5, 10])
a = unit(rng.standard_normal((50, 16))).astype(np.float32)
cases["identity"] = (a, a.copy(), [1, 5, 10])
cases["non_unit_gallery"] = (np.array([[1.0, 0.0], [3.0, 0.6]], dtype=np.float32),
                             np.array([[0.9, 0.3], [3.0, 0.6]], dtype=np.float32), [1])
t = np.zeros((4, 3), dtype=np.float32); t[:, 0] = 1.0
cases["exact_ties"] = (t, t.copy(), [1, 2])
a = unit(rng.standard_normal((10, 8))).astype(np.float32)
cases["fewer_queries_than_gallery"] = (a, a[:4].copy(), [1])
a, b = planted(40, 64, (1, 3, 11, 12), 5)
cases["configs_train_metric_k_1_10"] = (a, b, [1, 10])     # configs/pretrained_clip.jsonc:35
cases["scalar_k"] = (a, b, 5)                              # k_vals=5 default (metric.py:104-107)

arrays, desc = {}, {}
for name, (fa, fb, k) in cases.items():
    m = ref_metric.RecallAtK("visual", "titles", k)
    got = m.compute(fa, fb)
    arrays[f"{name}.a"], arrays[f"{name}.b"] = fa, fb
    arrays[f"{name}.recall"] = np.array([r for _, r in got], dtype=np.float64)
    desc[name] = {"k_vals": k, "ks_returned": [int(kk) for kk, _ in got]}
    # update()/result() path on the same case, two uneven batches (only defined when both sides have equal length)
    if fa.shape[0] == fb.shape[0] and fa.shape[0] >= 4:
        m.reset()
        cut = fa.shape[0] // 3
        for lo, hi in ((0, cut), (cut, fa.shape[0])):
            m.update(None, (torch.from_numpy(fa[lo:hi]), torch.from_numpy(fb[lo:hi])), None)
        res = m.result()
        desc[name]["result_keys"] = list(res)
        arrays[f"{name}.result"] = np.array(list(res.values()), dtype=np.float64)
np.savez_compressed(os.path.join(HERE, "recall_cases.npz"), case=np.array(json.dumps(desc)), **arrays)
print("wrote tests/golden/recall_cases.npz:", json.dumps(desc))
