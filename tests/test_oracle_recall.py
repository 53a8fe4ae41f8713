"""CPU: constructed rank cases for the RecallAtK restatement (model/metric.py:137-161)."""
import numpy as np
import torch

from oracle import eval_ref as E


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def planted(n=64, d=32, ranks=(1, 2, 5, 6, 10, 11), seed=0):
    """Gallery a = orthonormal-ish random unit rows; query i is built so that its own row i sits
    at a chosen rank: r-1 other gallery rows are strictly closer."""
    rng = np.random.default_rng(seed)
    a = unit(rng.standard_normal((n, d))).astype(np.float32)
    b = np.empty_like(a)
    want = np.array([ranks[i % len(ranks)] for i in range(n)])
    for i in range(n):
        others = [j for j in range(n) if j != i][: want[i] - 1]
        q = a[i].copy()
        # move the query toward `others` so they become closer than the target
        for j in others:
            q = q + 1.5 * a[j]
        b[i] = q
    return a, b, want


def test_permuted_identity_is_recall_one():
    rng = np.random.default_rng(1)
    a = unit(rng.standard_normal((50, 16))).astype(np.float32)
    res = E.recall_at_k(a, a.copy(), [1, 5, 10])
    assert res == [(1, 1.0), (5, 1.0), (10, 1.0)]


def test_planted_ranks_match_fp64_truth():
    a, b, want = planted()
    ids64, _ = E.l2_topk(a, b, 11, np.float64)
    rank = np.array([int(np.nonzero(ids64[i] == i)[0][0]) + 1 if (ids64[i] == i).any() else 99 for i in range(len(a))])
    got = dict(E.recall_at_k(a, b, [1, 5, 10]))
    for k in (1, 5, 10):
        assert abs(got[k] - (rank <= k).mean()) < 1e-12
    assert E.near_ties(a, b) == 0
    # depth is max(k)+1 and the hit test is `target in rp[:k]` (metric.py:145,153)
    assert 0 < got[1] < got[5] < got[10] < 1.0


def test_non_unit_gallery_l2_differs_from_cosine():
    """retrieval_evaluation.py:254-259 averages chunk embeddings WITHOUT renormalising, so the
    gallery is not unit-norm and squared-L2 order != cosine order (SURVEY 7 hard parts)."""
    a = np.array([[1.0, 0.0], [3.0, 0.6]], dtype=np.float32)   # row 1 has the larger cosine to q, but is far
    q = np.array([[0.9, 0.3], [3.0, 0.6]], dtype=np.float32)
    ids, _ = E.l2_topk(a, q, 2)
    cos = (q[0] @ a.T) / np.linalg.norm(a, axis=1)
    assert np.argmax(cos) == 1 and ids[0, 0] == 0


def test_exact_ties_lowest_index_first():
    a = np.zeros((4, 3), dtype=np.float32)
    a[:, 0] = 1.0                                              # four identical gallery rows
    ids, ds = E.l2_topk(a, a[:2], 4)
    assert ids.tolist() == [[0, 1, 2, 3], [0, 1, 2, 3]]
    # query 1's own row is at rank 2 under the lowest-index rule
    assert dict(E.recall_at_k(a, a.copy(), [1, 2]))[1] == 0.25


def test_denominator_is_gallery_size():
    rng = np.random.default_rng(3)
    a = unit(rng.standard_normal((10, 8))).astype(np.float32)
    b = a[:4].copy()                                           # fewer queries than gallery rows
    assert E.recall_at_k(a, b, [1])[0][1] == 4 / 10            # metric.py:138,158


def test_chunking_and_mean():
    fr = torch.arange(1 * 200 * 3 * 2 * 2, dtype=torch.float32).reshape(1, 200, 3, 2, 2)
    ch = E.chunk_frames(fr, frame_stride=16, nframes=8)        # 13 strided frames -> 8 + 5(resampled to 8)
    assert ch.shape == (2, 8, 3, 2, 2)
    strided = fr[:, ::16]
    assert torch.equal(ch[0], strided[0, :8])
    idx = torch.floor(torch.linspace(0, 4, 8)).long()
    assert torch.equal(ch[1], strided[0, 8:][idx])
    m = E.mean_chunks([torch.tensor([[1.0, 0.0], [0.0, 1.0]]), torch.tensor([[2.0, 2.0]])])
    assert torch.allclose(m, torch.tensor([[0.5, 0.5], [2.0, 2.0]]))   # not renormalised


def test_oracle_recall_equals_the_references_own_recallatk():
    """tests/golden/recall_cases.npz = outputs of the reference's own ``RecallAtK.compute`` / ``update`` / ``result``
    (model/metric.py:103-187, run by tests/golden/make_recall_golden.py under a numpy ``faiss`` stand-in): pins the
    oracle's bookkeeping -- depth max(k)+1, ``target in rp[:k]``, denominator len(features_a), direction naming."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "recall_cases.npz"), allow_pickle=False)
    desc = json.loads(str(z["case"]))
    assert len(desc) >= 7
    for name, c in desc.items():
        a, b = z[f"{name}.a"], z[f"{name}.b"]
        ks = c["k_vals"] if isinstance(c["k_vals"], list) else [c["k_vals"]]
        assert c["ks_returned"] == ks
        got = E.recall_at_k(a, b, ks)
        assert [k for k, _ in got] == ks
        assert np.array_equal(np.array([r for _, r in got]), z[f"{name}.recall"]), name
        if "result_keys" in c:
            both = E.recall_at_k(a, b, ks) + E.recall_at_k(b, a, ks)
            assert np.array_equal(np.array([r for _, r in both]), z[f"{name}.result"]), name
            assert c["result_keys"] == [f"titles_from_visual-recall_at_{k}" for k in ks] + \
                [f"visual_from_titles-recall_at_{k}" for k in ks]


def test_certificate_error_bound_needs_the_full_bf16_roundoff():
    """The EXACT sweep's certificate (vtc_amd/csrc/sweep.hip minsel_kernel, restated in oracle/sweep_planes.certificate_sets)
    is only as good as its per-entry error bound.  On the midpoint case the constant of rounds 1-2 (bf16 unit roundoff taken
    as 2^-9) certifies a candidate list WITHOUT the true nearest row; the corrected one (2^-8) keeps it.  Pins the reasoning
    behind exact2_kappa on the CPU; the kernel itself is checked against fp64 brute force in tests/test_gpu_sweep.py."""
    from oracle import sweep_planes as SP
    d, depth = 512, 11
    g, q = SP.midpoint_case(d=d, depth=depth)
    exact = ((g.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    assert int(np.argmin(exact)) == 0
    k_old = 2.0 ** -8 * (1 + 2.0 ** -10) + 2.0 * d / 2 ** 24 + 2.0 ** -16 + 1e-6
    k_new = 2.0 ** -7 * (1 + 2.0 ** -9) + 2.0 * d / 2 ** 24 + 2.0 ** -15 + 1e-6
    cand_old, cert_old, approx = SP.certificate_sets(g, q, depth, k_old)
    cand_new, cert_new, _ = SP.certificate_sets(g, q, depth, k_new)
    assert cert_old and 0 not in cand_old                     # "certified", and wrong
    assert cert_new and 0 in cand_new
    err = approx.astype(np.float64) - exact
    bound = k_new * (float((q.astype(np.float64) ** 2).sum()) + float((g.astype(np.float64) ** 2).sum(1).max()))
    assert np.abs(err).max() <= bound and np.abs(err).max() > 0.5 * bound * 0.55     # the case really exercises the bound


def test_measured_rounding_error_bound_is_rigorous_and_tighter():
    """Round 4: the certificate's eps is built from the rows' MEASURED bf16 rounding errors (|x - bf16(x)|, exact in fp32) instead
    of the worst case 2^-8 |x| (oracle/sweep_planes.measured_eps = sweep.hip sweep_prep_kernel + minsel_kernel).  It must (a) still
    bound every entry's error -- also on the adversarial midpoint case, where every coordinate's error IS the worst case, so the
    bound may not shrink there -- and keep the true nearest row in the certified list; (b) be several times smaller on ordinary
    embeddings, which is what shrinks the candidate sets."""
    from oracle import sweep_planes as SP
    d, depth = 512, 11
    k_fp32 = 2.0 * d / 2 ** 24 + 2.0 ** -15 + 1e-6
    k_worst = 2.0 ** -7 * (1 + 2.0 ** -9) + k_fp32
    # (a) adversarial
    g, q = SP.midpoint_case(d=d, depth=depth)
    exact = ((g.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    eps = SP.measured_eps(g, q, k_fp32)
    cand, cert, approx = SP.certificate_sets(g, q, depth, None, eps=eps)
    assert np.abs(approx.astype(np.float64) - exact).max() <= eps
    assert cert and 0 in cand
    worst = k_worst * (float((q.astype(np.float64) ** 2).sum()) + float((g.astype(np.float64) ** 2).sum(1).max()))
    assert eps > 0.45 * worst                # the rows that carry the case sit on midpoints: no free lunch there (the +-1 fillers round exactly)
    # (b) unit-norm random embeddings with a planted neighbourhood
    rng = np.random.default_rng(3)
    g = rng.standard_normal((4096, d)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    q = (g[7] + 0.05 * rng.standard_normal(d).astype(np.float32))[None]
    q /= np.linalg.norm(q)
    exact = ((g.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    eps = SP.measured_eps(g, q, k_fp32)
    worst = k_worst * (float((q.astype(np.float64) ** 2).sum()) + float((g.astype(np.float64) ** 2).sum(1).max()))
    cand_m, cert_m, approx = SP.certificate_sets(g, q, depth, None, eps=eps)
    cand_w, cert_w, _ = SP.certificate_sets(g, q, depth, k_worst)
    assert np.abs(approx.astype(np.float64) - exact).max() <= eps
    assert eps < 0.5 * worst and cand_m.size <= cand_w.size      # measured |e| ~ 0.0016 |x| on random data against the worst case 0.0039 |x|
    top = np.argsort(exact, kind="stable")[:depth]
    assert cert_m and set(top) <= set(cand_m.tolist())
