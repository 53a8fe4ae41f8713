"""INTEGRATION.md section B shows the ctypes bindings a maintainer of the reference would add (model/metric.py:140-146 -> vtc_l2_topk,
:177-180 -> vtc_l2_recall_bidir).  The two python blocks are EXECUTED here as written, against the built library, and held to the fp64 oracle:
a binding in the document that does not run is worse than none."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import eval_ref as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return [b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "_lib" in b and "def " in b]


@pytest.mark.gpu
def test_the_documented_reference_side_bindings_run_and_agree_with_the_oracle():
    from vtc_amd import _lib as L
    blocks = _blocks()
    assert len(blocks) == 2, len(blocks)
    ns = {}
    for b in blocks:
        exec(b.replace('ctypes.CDLL("libvtc_hip.so")', f'ctypes.CDLL("{L.LIB_PATH}")'), ns)
    rng = np.random.default_rng(3)
    n, d = 1536, 512
    a = rng.standard_normal((n, d)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
    ids = ns["knn_ids"](a, b, 11)
    assert np.array_equal(ids, E.l2_topk(a, b, 11, np.float64)[0])
    r_ab, r_ba = ns["recall_both"](torch.from_numpy(a), torch.from_numpy(b))
    assert r_ab == dict(E.recall_at_k(a, b, [1, 5, 10], np.float64)) and r_ba == dict(E.recall_at_k(b, a, [1, 5, 10], np.float64))
