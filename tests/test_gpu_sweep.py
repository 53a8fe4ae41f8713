"""GPU parity of the N x N sweep, similarity and loss against the oracle (oracle/eval_ref.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import eval_ref as E
from oracle import model_ref as M

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def planted(n, d, seed, noise=0.6):
    rng = np.random.default_rng(seed)
    a = unit(rng.standard_normal((n, d))).astype(np.float32)
    b = unit(a + noise * unit(rng.standard_normal((n, d))) * rng.uniform(0.2, 1.8, (n, 1))).astype(np.float32)
    return a, b


@pytest.mark.parametrize("n,d", [(300, 64), (1000, 512), (4099, 512)])
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_l2_topk_matches_oracle(n, d, prec):
    from vtc_amd import _lib as L
    from vtc_amd import ops
    a, b = planted(n, d, seed=n)
    depth = 11
    ids, dists = ops.l2_topk(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), depth,
                             precision=L.SWEEP_F32 if prec == "f32" else L.SWEEP_BF16X3, rows_per_block=256)
    ids, dists = ids.cpu().numpy(), dists.cpu().numpy()
    ids64, d64 = E.l2_topk(a, b, depth, np.float64)
    # fp32: rounding of |q|^2+|g|^2-2q.g at magnitude ~2 (ulp 2.4e-7); bf16x3: 2 x the split-product error,
    # which scales with the element size 1/sqrt(d)
    tol = 4e-6 if prec == "f32" else 2e-5 * (64 / d) ** 0.5
    assert np.abs(dists - d64).max() < tol
    # ranks identical wherever the fp64 gaps are not within rounding
    gaps = np.diff(E.l2_topk(a, b, depth + 1, np.float64)[1], axis=1)
    safe = (gaps > 4 * tol).all(axis=1)
    assert safe.mean() > 0.9
    assert np.array_equal(ids[safe], ids64[safe])
    # R@K identical to the oracle's literal restatement of metric.py:137-161
    hits = ops.recall_hits(torch.from_numpy(ids).cuda(), [1, 5, 10]).cpu().numpy()
    # against the GROUND-TRUTH (fp64) ranks: an approximate mode can only differ on a query whose R@K outcome hangs on a
    # gap below its resolution -- the count of those is reported and is the bound (no silent skip); the
    # default EXACT mode must equal them outright
    ref = E.recall_at_k(a, b, [1, 5, 10], np.float64)
    nt = E.near_ties(a, b, tol=4 * tol)
    print(f"[parity] l2_topk {prec} n={n}: near ties (fp64 gap < {4 * tol:.1e}) = {nt}")
    assert all(abs(int(h) - round(r * n)) <= nt for h, (_, r) in zip(hits, ref))
    ids_x, _ = ops.l2_topk(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), depth, precision=L.SWEEP_EXACT)
    assert np.array_equal(ids_x.cpu().numpy(), ids64)
    hits_x = ops.recall_hits(ids_x, [1, 5, 10]).cpu().numpy()
    assert [h / n for h in hits_x] == [r for _, r in ref]


def test_recall_metric_dropin_and_ties():
    from vtc_amd.host.metric import RecallAtK
    a, b = planted(777, 128, seed=5)
    got = RecallAtK("videos", "titles", [1, 5, 10]).compute(a, b)
    assert got == E.recall_at_k(a, b, [1, 5, 10])
    # exact ties resolve to the lowest gallery index
    t = np.zeros((8, 64), dtype=np.float32); t[:, 0] = 1.0
    m = RecallAtK("a", "b", [1, 2])
    assert m.topk_ids(t, t[:3]).cpu().numpy().tolist() == [[0, 1, 2]] * 3
    # non-unit gallery (mean of chunk embeddings is not renormalised): L2 != cosine
    g = np.zeros((2, 64), dtype=np.float32); g[0, 0] = 1.0; g[1, 0] = 3.0; g[1, 1] = 0.6
    q = np.zeros((1, 64), dtype=np.float32); q[0, 0] = 0.9; q[0, 1] = 0.3
    assert m.topk_ids(g, q).cpu().numpy()[0, 0] == 0
    # ragged / small: depth clamps to the gallery size, k larger than gallery
    s = RecallAtK("a", "b", [1, 5, 10]).compute(a[:4], b[:4])
    assert s == E.recall_at_k(a[:4], b[:4], [1, 5, 10])


def test_similarity_and_clip_loss():
    from vtc_amd import ops
    case, g = load_golden("clip_loss.npz")
    for i in range(3):
        v = float(ops.clip_loss(torch.from_numpy(g[f"sim{i}"]).cuda()))
        assert abs(v - g["loss"][i]) < 2e-6
    rng = np.random.default_rng(0)
    v, t = unit(rng.standard_normal((50, 512))).astype(np.float32), unit(rng.standard_normal((37, 512))).astype(np.float32)
    ls = torch.tensor(float(np.log(1 / 0.07)))
    sim = ops.similarity(torch.from_numpy(v).cuda(), torch.from_numpy(t).cuda(), ls.cuda()).cpu().numpy()
    ref = np.exp(np.float64(ls)) * (v.astype(np.float64) @ t.astype(np.float64).T)
    assert np.abs(sim - ref).max() < 1e-5
    n = 256
    s = torch.randn(n, n) * 4
    assert abs(float(ops.clip_loss(s.cuda())) - float(M.clip_loss(s))) < 1e-5


@pytest.mark.parametrize("prec", ["exact", "f32"])
def test_full_size_properties_10k(prec):
    """BASELINE size (10k x 10k): size-independent properties instead of a full oracle run:
    self-search finds itself at rank 1 with distance ~0, dists ascending, ids a valid set,
    and a row sample agrees with the fp64 oracle."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    p = L.SWEEP_EXACT if prec == "exact" else L.SWEEP_F32
    n, d = 10000, 512
    a, b = planted(n, d, seed=1)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ids, dists = ops.l2_topk(ta, ta, 11, precision=p)
    assert torch.equal(ids[:, 0].cpu(), torch.arange(n))
    assert dists[:, 0].abs().max() < 1e-5
    ids, dists = ops.l2_topk(ta, tb, 11, precision=p)
    dn = dists.cpu().numpy()
    assert (np.diff(dn, axis=1) >= 0).all()
    idn = ids.cpu().numpy()
    assert idn.min() >= 0 and idn.max() < n and all(len(set(r)) == 11 for r in idn[::97])
    rows = np.arange(0, n, 41)
    i64, d64 = E.l2_topk(a, b[rows], 11, np.float64)
    assert np.abs(dn[rows] - d64).max() < 4e-6
    assert (idn[rows] == i64).mean() > 0.999
    if prec == "exact":
        assert np.array_equal(idn[rows], i64)


@pytest.mark.parametrize("n,d", [(300, 64), (1000, 512), (4099, 512), (1500, 768)])      # 768: ViT-L/14's embedding width
def test_l2_topk_exact_mode_has_fp64_ranks(n, d):
    """VTC_SWEEP_EXACT: BF16X3 candidate lists re-ranked in fp64 -> the ids are those of exact fp64 arithmetic on
    EVERY row (no near-tie exemption), the distances are the fp64 distances rounded to fp32."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    a, b = planted(n, d, seed=n + 1)
    for depth in (1, 11, 40):
        ids, dists = ops.l2_topk(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), depth, precision=L.SWEEP_EXACT)
        ids64, d64 = E.l2_topk(a, b, depth, np.float64)
        direct = ((b[:, None, :].astype(np.float64) - a[ids64].astype(np.float64)) ** 2).sum(-1)   # sum (q - g)^2
        assert np.array_equal(ids.cpu().numpy(), ids64)
        assert np.abs(dists.cpu().numpy() - direct).max() < 3e-7


def test_l2_topk_exact_mode_dense_near_ties_take_the_fp64_brute_force():
    """A cluster of near-duplicates spaced far below the split-bf16 resolution, exact duplicates, and a gallery
    smaller than the candidate list: rows that cannot be certified are recomputed by fp64 brute force."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    rng = np.random.default_rng(9)
    a = unit(rng.standard_normal((2000, 512))).astype(np.float32)
    base = a[7].copy()
    for j in range(60):                                      # 60 near-duplicates of row 7, 1e-7-scale differences
        a[100 + j] = base
        a[100 + j, j % 512] += np.float32(1e-7 * (j + 1))
    a[500:530] = a[499]                                      # 30 exact duplicates: ties resolve to the lowest index
    q = np.stack([base, a[499], unit(rng.standard_normal(512)).astype(np.float32)])
    ids, _ = ops.l2_topk(torch.from_numpy(a).cuda(), torch.from_numpy(q).cuda(), 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(ids.cpu().numpy(), E.l2_topk(a, q, 11, np.float64)[0])
    small = a[:20]                                           # gallery smaller than the candidate depth
    ids, _ = ops.l2_topk(torch.from_numpy(small).cuda(), torch.from_numpy(q).cuda(), 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(ids.cpu().numpy(), E.l2_topk(small, q, 11, np.float64)[0])


def _bidir_cases():
    out = []
    for prec in ("exact", "f32", "bf16x3"):
        for na, nb, d in ((300, 300, 64), (1000, 777, 512), (4099, 4099, 512), (40, 3000, 128), (3000, 40, 128)):
            # the approximate modes' 4099-row case spends 13 s each in the fp64 oracle's gap scan: extended
            marks = [pytest.mark.extended] if (na == 4099 and prec != "exact") else []
            out.append(pytest.param(na, nb, d, prec, marks=marks, id=f"{prec}-{na}-{nb}-{d}"))
    return out


@pytest.mark.parametrize("na,nb,d,prec", _bidir_cases())
def test_l2_topk_bidir_equals_two_searches(na, nb, d, prec):
    """vtc_l2_topk_bidir reads the second direction off the columns of the first direction's distance blocks: it
    must return what two vtc_l2_topk calls return (EXACT: the fp64 ids on every row of both directions)."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    p = {"exact": L.SWEEP_EXACT, "f32": L.SWEEP_F32, "bf16x3": L.SWEEP_BF16X3}[prec]
    rng = np.random.default_rng(na + nb)
    a = unit(rng.standard_normal((na, d))).astype(np.float32)
    b = unit(rng.standard_normal((nb, d))).astype(np.float32)
    m = min(na, nb)
    b[:m] = unit(a[:m] + 0.5 * unit(rng.standard_normal((m, d)))).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for depth in (1, 11):
        for rpb in (0, 256):                                   # one block / several blocks (partial column lists merged)
            i1, d1, i2, d2 = ops.l2_topk_bidir(ta, tb, depth, precision=p, rows_per_block=rpb)
            r1, e1 = E.l2_topk(a, b, depth, np.float64)           # gallery a, queries b
            r2, e2 = E.l2_topk(b, a, depth, np.float64)           # gallery b, queries a
            tol = 4e-6 if prec != "bf16x3" else 2e-5 * (64 / d) ** 0.5
            assert np.abs(d1.cpu().numpy() - e1).max() < tol and np.abs(d2.cpu().numpy() - e2).max() < tol
            if prec == "exact":
                assert np.array_equal(i1.cpu().numpy(), r1) and np.array_equal(i2.cpu().numpy(), r2)
            else:
                for got, ref, g, q in ((i1, r1, a, b), (i2, r2, b, a)):
                    gaps = np.diff(E.l2_topk(g, q, depth + 1, np.float64)[1], axis=1)
                    safe = (gaps > 4 * tol).all(axis=1)
                    assert safe.mean() > 0.9 and np.array_equal(got.cpu().numpy()[safe], ref[safe])


def test_l2_topk_bidir_ties_and_duplicates():
    """Exact duplicates on both sides: (distance, index) order in both directions, lowest index first."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    rng = np.random.default_rng(3)
    a = unit(rng.standard_normal((500, 128))).astype(np.float32)
    b = unit(rng.standard_normal((400, 128))).astype(np.float32)
    a[100:130] = a[99]
    b[200:240] = b[7]
    for p in (L.SWEEP_EXACT, L.SWEEP_F32):
        i1, _, i2, _ = ops.l2_topk_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 11, precision=p, rows_per_block=128)
        j1, _ = ops.l2_topk(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 11, precision=p)
        j2, _ = ops.l2_topk(torch.from_numpy(b).cuda(), torch.from_numpy(a).cuda(), 11, precision=p)
        if p == L.SWEEP_EXACT:
            assert torch.equal(i1, j1) and torch.equal(i2, j2)
            assert np.array_equal(i2.cpu().numpy(), E.l2_topk(b, a, 11, np.float64)[0])
        else:
            assert torch.equal(i1, j1)
            # fp32: the transposed direction rounds |b|^2 - 2 a.b + |a|^2 in another order; duplicates still tie exactly
            assert (i2 == j2).float().mean() > 0.99


def test_recall_compute_both_matches_two_computes_and_oracle():
    """RecallAtK.compute_both on the one-matrix path (threshold lowered) == the reference's two compute() calls."""
    from vtc_amd.host.metric import RecallAtK
    a, b = planted(1500, 128, seed=11)
    m = RecallAtK("videos", "titles", [1, 5, 10])
    two = (m.compute(a, b), m.compute(b, a))
    m.bidir_min_rows = 0
    assert m.compute_both(a, b) == two
    assert two[0] == E.recall_at_k(a, b, [1, 5, 10]) and two[1] == E.recall_at_k(b, a, [1, 5, 10])


@pytest.mark.parametrize("prec", ["exact", pytest.param("f32", marks=pytest.mark.extended), pytest.param("bf16x3", marks=pytest.mark.extended)])
def test_stress_size_50k_bidir_equals_two_searches_and_oracle_sample(prec):
    """BASELINE configs[4]: the 50k x 50k sweep (D = 512, depth 11).  At this size the distance matrix is walked in
    five 2 GiB row blocks, the column lists are carried from block to block and merged over segments -- none of which
    the <= 10k cases reach.  Properties: (i) vtc_l2_topk_bidir == two vtc_l2_topk calls on every row (EXACT: identical
    ids; F32 / BF16X3: identical wherever the fp64 gap to the next neighbour is resolvable), (ii) a >= 1000-row sample
    per direction == the fp64 oracle, (iii) R@1/5/10 from the ids == the oracle's hit rule on the sample."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    p = {"exact": L.SWEEP_EXACT, "f32": L.SWEEP_F32, "bf16x3": L.SWEEP_BF16X3}[prec]
    n, d, depth = 50000, 512, 11
    a, b = planted(n, d, seed=50)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    i1, d1, i2, d2 = ops.l2_topk_bidir(ta, tb, depth, precision=p)
    j1, e1 = ops.l2_topk(ta, tb, depth, precision=p)          # gallery a, queries b
    j2, e2 = ops.l2_topk(tb, ta, depth, precision=p)          # gallery b, queries a
    torch.cuda.synchronize()
    i1n, i2n, j1n, j2n = (t.cpu().numpy() for t in (i1, i2, j1, j2))
    assert i1n.min() >= 0 and i1n.max() < n and i2n.min() >= 0 and i2n.max() < n
    assert (np.diff(d1.cpu().numpy(), axis=1) >= 0).all() and (np.diff(d2.cpu().numpy(), axis=1) >= 0).all()
    if prec == "exact":
        assert np.array_equal(i1n, j1n) and np.array_equal(i2n, j2n)
    else:
        assert (i1n == j1n).mean() > 0.9995 and (i2n == j2n).mean() > 0.999
    rows = np.arange(7, n, 47)[:1024]
    assert len(rows) >= 1000
    for got, gal, qry in ((i1n, a, b), (i2n, b, a)):
        ref_ids, ref_d = E.l2_topk(gal, qry[rows], depth + 1, np.float64)
        if prec == "exact":
            assert np.array_equal(got[rows], ref_ids[:, :depth])
        else:
            safe = (np.diff(ref_d, axis=1) > 2e-5).all(axis=1)
            assert safe.mean() > 0.9 and np.array_equal(got[rows][safe], ref_ids[safe][:, :depth])
        # the hit rule of model/metric.py:148-160 on the sample rows (target = the row's own index)
        for k in (1, 5, 10):
            want = sum(int(rows[r] in ref_ids[r, :k]) for r in range(len(rows)))
            have = sum(int(rows[r] in got[rows[r], :k]) for r in range(len(rows)))
            if prec == "exact":
                assert have == want
            else:
                assert abs(have - want) <= 2
    # the whole-matrix R@K through the metric object (one-matrix path at this size) == counting hits on the ids above
    from vtc_amd.host.metric import RecallAtK
    m = RecallAtK("videos", "titles", [1, 5, 10])
    m.precision = p
    r_ab, r_ba = m.compute_both(ta, tb)
    tgt = np.arange(n)[:, None]
    for (k, r), idsn in list(zip(r_ab, [i1n] * 3)) + list(zip(r_ba, [i2n] * 3)):
        assert abs(r - float((idsn[:, :k] == tgt).any(axis=1).mean())) < 1e-12


@pytest.mark.parametrize("scale", [1.0, 25.0])
def test_exact_block_minima_path_on_unnormalised_and_clustered_data(scale):
    """The block-minima EXACT path (galleries of >= 1024 rows): embeddings of very different norms (the error bound of the
    bf16 distance keys scales with |q|^2 + max|g|^2), tight clusters (many rows inside one 2-eps window and inside one
    64-row block: the rows the certificate must refuse and hand to the fp64 brute force), self-search (distances that
    round negative clamp to zero).  ids == fp64 oracle on EVERY row, both directions, one-matrix and two-search forms."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    rng = np.random.default_rng(77)
    n, d = 3000, 128
    a = rng.standard_normal((n, d)).astype(np.float32) * rng.uniform(0.2, 1.0, (n, 1)).astype(np.float32) * np.float32(scale)
    centres = a[rng.integers(0, 40, n)]                      # 40 clusters ...
    a[500:1500] = centres[500:1500] + np.float32(0.02 * scale) * rng.standard_normal((1000, d)).astype(np.float32)
    a[64:128] = a[64] + np.float32(1e-3 * scale) * rng.standard_normal((64, d)).astype(np.float32)   # ... and one whole block of near-duplicates
    b = a + np.float32(0.05 * scale) * rng.standard_normal((n, d)).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    r1 = E.l2_topk(a, b, 11, np.float64)[0]
    r2 = E.l2_topk(b, a, 11, np.float64)[0]
    i1, _, i2, _ = ops.l2_topk_bidir(ta, tb, 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(i1.cpu().numpy(), r1) and np.array_equal(i2.cpu().numpy(), r2)
    j1, _ = ops.l2_topk(ta, tb, 11, precision=L.SWEEP_EXACT)
    j2, _ = ops.l2_topk(tb, ta, 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(j1.cpu().numpy(), r1) and np.array_equal(j2.cpu().numpy(), r2)
    s, _ = ops.l2_topk(ta, ta, 5, precision=L.SWEEP_EXACT)
    assert np.array_equal(s.cpu().numpy(), E.l2_topk(a, a, 5, np.float64)[0])


@pytest.mark.parametrize("n,world,d,scale", [(4099, 3, 512, 1.0), (10000, 8, 512, 1.0), (6000, 2, 128, 25.0),
                                              pytest.param(50000, 8, 512, 1.0, marks=pytest.mark.extended)])     # (50k at 8 ranks in the default run: the recall-only form below)
def test_sharded_one_matrix_sweep_equals_single_gpu_search(n, world, d, scale):
    """The one-GEMM-per-rank sweep (vtc_l2_sweep_shard_rows -> exchange -> vtc_l2_sweep_shard_cols, include/vtc_hip.h),
    the ranks played one after the other on this card with the exchange of vtc_amd/dist.py done by slicing: ids of BOTH
    directions bit-identical to the single-GPU EXACT search (which test_stress_size... pins to the fp64 oracle)."""
    from vtc_amd import _lib as L
    from vtc_amd import dist as vdist
    from vtc_amd import ops
    a, b = planted(n, d, seed=n + world)
    if scale != 1.0:       # un-normalised, clustered: norms differ by rows, so the per-column error bound matters
        rng = np.random.default_rng(5)
        a = (a * rng.uniform(0.5, scale, (n, 1))).astype(np.float32)
        b = (b * rng.uniform(0.5, scale, (n, 1))).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    depth = 11
    bounds = [vdist.shard_bounds(n, r, world) for r in range(world)]
    assert all(ops.sweep_shard_supported(n, hi - lo, depth) for lo, hi in bounds)
    rb = ops.sweep_row_block()
    nbp = -(-max(hi - lo for lo, hi in bounds) // rb)
    rows, planes = [], []
    for lo, hi in bounds:
        i1, pl = ops.sweep_shard_rows(ta, tb[lo:hi], depth, nbp)
        rows.append(i1)
        planes.append(pl)
    src_base = torch.tensor([lo for lo, _ in bounds], dtype=torch.int32, device="cuda")
    cols = []
    for lo, hi in bounds:
        recv = torch.stack([pl[:, :, lo:hi] for pl in planes]).contiguous()          # what the all-to-all delivers to this rank
        cols.append(ops.sweep_shard_cols(tb, ta[lo:hi], depth, recv, src_base))
    ref_b2a = ops.l2_topk(ta, tb, depth, precision=L.SWEEP_EXACT, return_dists=False)[0]
    ref_a2b = ops.l2_topk(tb, ta, depth, precision=L.SWEEP_EXACT, return_dists=False)[0]
    assert torch.equal(torch.cat(rows), ref_b2a)
    assert torch.equal(torch.cat(cols), ref_a2b)


def test_sharded_one_matrix_sweep_random_shapes():
    """Seeded random (n, world, d, depth): ragged shards, padded blocks, odd row counts (scalar plane loads), depth 1 .. 32 --
    both directions bit-identical to the single-GPU EXACT search."""
    from vtc_amd import _lib as L
    from vtc_amd import dist as vdist
    from vtc_amd import ops
    rng = np.random.default_rng(2026)
    for case in range(8):
        n = int(rng.integers(1100, 9000))
        world = int(rng.integers(2, 7))
        d = int(rng.choice([64, 128, 512]))
        depth = int(rng.choice([1, 6, 11, 32]))
        bounds = [vdist.shard_bounds(n, r, world) for r in range(world)]
        if not all(ops.sweep_shard_supported(n, hi - lo, depth) for lo, hi in bounds):
            continue
        a, b = planted(n, d, seed=1000 + case)
        if case % 3 == 2:                                     # duplicates: exact ties, lowest index first
            a[n // 2:n // 2 + 40] = a[:40]
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        rb = ops.sweep_row_block()
        nbp = -(-max(hi - lo for lo, hi in bounds) // rb)
        rows, planes = zip(*[ops.sweep_shard_rows(ta, tb[lo:hi], depth, nbp) for lo, hi in bounds])
        src_base = torch.tensor([lo for lo, _ in bounds], dtype=torch.int32, device="cuda")
        cols = [ops.sweep_shard_cols(tb, ta[lo:hi], depth, torch.stack([pl[:, :, lo:hi] for pl in planes]).contiguous(), src_base)
                for lo, hi in bounds]
        ref_b2a = ops.l2_topk(ta, tb, depth, precision=L.SWEEP_EXACT, return_dists=False)[0]
        ref_a2b = ops.l2_topk(tb, ta, depth, precision=L.SWEEP_EXACT, return_dists=False)[0]
        assert torch.equal(torch.cat(rows), ref_b2a), (n, world, d, depth)
        assert torch.equal(torch.cat(cols), ref_a2b), (n, world, d, depth)
        # and against the fp64 oracle on a sample of rows
        idx = rng.choice(n, size=64, replace=False)
        ids64, _ = E.l2_topk(b, a[idx], depth, np.float64)
        assert np.array_equal(torch.cat(cols).cpu().numpy()[idx], ids64), (n, world, d, depth)


def test_exact_sweep_certificate_holds_on_coordinated_bf16_midpoints():
    """the block-minima certificate needs eps >= |approx - exact| per entry, and bf16's unit roundoff is
    2^-8 -- rounds 1-2 used 2^-9.  oracle.sweep_planes.midpoint_case puts every coordinate on a round-to-even midpoint so that
    the operand roundings add up (+4.6 on the true nearest row, -4.6 on the eleven rows just behind it): with the halved
    constant the 'certified' list misses the true #1 (tests/test_oracle_recall.py shows that on the CPU restatement); the ids
    must be those of fp64 brute force -- single search, both directions from one matrix, and the split-bf16 path (gallery
    below 1 024 rows)."""
    from oracle import sweep_planes as SP
    from vtc_amd import _lib as L
    from vtc_amd import ops
    for n in (2048, 832):
        g, q = SP.midpoint_case(n_gallery=n)
        # more queries around the adversarial one (row 0): small perturbations of it on exactly representable steps
        rng = np.random.default_rng(3)
        qs = np.repeat(q, 40, 0)
        qs[1:] += (rng.integers(-2, 3, size=qs[1:].shape) * 2.0 ** -6).astype(np.float32)
        ids, _ = ops.l2_topk(torch.from_numpy(g).cuda(), torch.from_numpy(qs).cuda(), 11, precision=L.SWEEP_EXACT)
        ids64, _ = E.l2_topk(g, qs, 11, np.float64)
        assert np.array_equal(ids.cpu().numpy(), ids64), n
        assert ids[0, 0].item() == 0
    g, q = SP.midpoint_case(n_gallery=2048)
    qs = np.concatenate([q, g[1:1100]])                      # >= 1 024 rows on both sides: the one-matrix path
    i1, _, i2, _ = ops.l2_topk_bidir(torch.from_numpy(g).cuda(), torch.from_numpy(qs).cuda(), 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(i1.cpu().numpy(), E.l2_topk(g, qs, 11, np.float64)[0])
    assert np.array_equal(i2.cpu().numpy(), E.l2_topk(qs, g, 11, np.float64)[0])


@pytest.mark.parametrize("n", [2000, 6000])
def test_sharded_recall_world_1_equals_recallatk_compute_both(n):
    """`bench.py --gpus 1` times vdist.sharded_recall(world=1); the single-GPU callers (evaluation/eval.py, RecallAtK.result)
    go through RecallAtK.compute_both.  Same R@K, both below and above the one-matrix threshold -- so an N = 1 SCALE line and the BENCH
    line measure the same computation -- and both equal the fp64 oracle."""
    from vtc_amd import dist as vdist
    from vtc_amd.host.metric import RecallAtK
    a, b = planted(n, 512, seed=n + 3)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ph = {}
    r_ab, r_ba = vdist.sharded_recall(ta, tb, n, [1, 5, 10], 0, 1, phases=ph)
    m = RecallAtK("videos", "titles", [1, 5, 10])
    c_ab, c_ba = m.compute_both(ta, tb)
    assert r_ab == dict(c_ab) and r_ba == dict(c_ba)
    assert r_ab == dict(E.recall_at_k(a, b, [1, 5, 10], np.float64)) and r_ba == dict(E.recall_at_k(b, a, [1, 5, 10], np.float64))
    one_matrix_from = min(vdist.BIDIR_MIN_ROWS, vdist.RANK_MIN_ROWS) if vdist.RANK_PATH else vdist.BIDIR_MIN_ROWS       # (VTC_SWEEP_RANK=0: sorted lists)
    assert ph["path"].startswith("one distance matrix" if n >= one_matrix_from else "two searches")
    assert vdist.RANK_MIN_ROWS == m.rank_min_rows
    assert vdist.BIDIR_MIN_ROWS == m.bidir_min_rows and vdist.BIDIR_MIN_ROWS_F32 == m.bidir_min_rows_f32


def test_exact_sweep_with_rows_whose_components_underflow_when_squared():
    """the EXACT certificate uses the rows' MEASURED bf16 rounding errors |x - bf16(x)|, accumulated as squares in fp32; a
    row whose components sit below ~1e-19 squares to flushed zeros, so its measured error (and norm) would read 0.  The prep kernel adds
    d x FLT_MIN under its roots; whatever the certificate then decides, the ids must still be the fp64 ones -- for tiny rows among
    ordinary ones, tiny queries, and an all-tiny problem (every key underflows: the fp64 brute force takes over)."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    rng = np.random.default_rng(21)
    a = unit(rng.standard_normal((600, 64))).astype(np.float32)
    b = unit(a + 0.3 * unit(rng.standard_normal((600, 64)))).astype(np.float32)
    a[100:180] *= np.float32(1e-18)
    b[50:90] *= np.float32(3e-19)
    for ga, qb in ((a, b), (b, a), ((a * np.float32(1e-17)).astype(np.float32), (b * np.float32(1e-17)).astype(np.float32))):
        ids, _ = ops.l2_topk(torch.from_numpy(ga).cuda(), torch.from_numpy(qb).cuda(), 11, precision=L.SWEEP_EXACT)
        assert np.array_equal(ids.cpu().numpy(), E.l2_topk(ga, qb, 11, np.float64)[0])
    i1, _, i2, _ = ops.l2_topk_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 11, precision=L.SWEEP_EXACT)
    assert np.array_equal(i1.cpu().numpy(), E.l2_topk(a, b, 11, np.float64)[0]) and np.array_equal(i2.cpu().numpy(), E.l2_topk(b, a, 11, np.float64)[0])


def _hits_ref(a, b, ks):
    n = a.shape[0]
    return np.array([[round(r * n) for _, r in E.recall_at_k(a, b, ks, np.float64)], [round(r * n) for _, r in E.recall_at_k(b, a, ks, np.float64)]])


@pytest.mark.parametrize("n,d,noise", [(1024, 64, 0.6), (2500, 512, 0.6), (4099, 512, 1.5), (3000, 128, 0.0), (2048, 768, 0.3)])
def test_recall_bidir_rank_path_equals_the_fp64_oracle(n, d, noise):
    """vtc_l2_recall_bidir (round 5): the hit counters of both directions straight from the distance GEMM's key planes -- the RANK of each
    query's own gallery row, no sorted neighbour lists -- must be the fp64 oracle's and the two-step form's (vtc_l2_topk_bidir +
    vtc_recall_hits_pair), for planted positives at several noise levels (noise 0: b == a, every rank 0 with exact ties), k sets incl. a
    single k, and for UNRELATED sets (every target far from the top: the certain-miss shortcut)."""
    from vtc_amd import _lib as L
    from vtc_amd import ops
    a, b = planted(n, d, seed=n, noise=noise) if noise > 0 else (planted(n, d, seed=n)[0],) * 2
    ta, tb = torch.from_numpy(np.ascontiguousarray(a)).cuda(), torch.from_numpy(np.ascontiguousarray(b)).cuda()
    for ks in ([1, 5, 10], [1], [3, 7, 20, 50]):
        got = ops.recall_bidir(ta, tb, ks).cpu().numpy()
        assert np.array_equal(got, _hits_ref(a, b, ks)), (n, d, noise, ks, got.tolist(), _hits_ref(a, b, ks).tolist())
        i1, _, i2, _ = ops.l2_topk_bidir(ta, tb, min(max(ks) + 1, 64), precision=L.SWEEP_EXACT, return_dists=False)
        two = ops.recall_hits_pair(i1, i2, [min(k, i1.shape[1]) for k in ks], 0, torch.zeros(2, len(ks), dtype=torch.int64, device="cuda")).cpu().numpy()
        if max(ks) < 64:
            assert np.array_equal(got, two)
    rng = np.random.default_rng(n)
    c = unit(rng.standard_normal((n, d))).astype(np.float32)            # unrelated to a: recall ~ k / n
    got = ops.recall_bidir(ta, torch.from_numpy(c).cuda(), [1, 5, 10]).cpu().numpy()
    assert np.array_equal(got, _hits_ref(a, c, [1, 5, 10]))


def test_recall_bidir_rank_path_ties_duplicates_scales_and_the_adversarial_case():
    """Exact duplicates in both sets (ties resolve to the lowest index: a duplicate BEFORE the target outranks it, one after does not), dense
    near-duplicate clusters around targets (the lists overflow: fp64 fallback), un-normalised rows at scale 25 (eps scales with the norms),
    and the bf16-midpoint adversary (rows whose rounding errors are the worst case: no entry may be dropped as certain that is not)."""
    from oracle import sweep_planes as SP
    from vtc_amd import ops
    rng = np.random.default_rng(5)
    n, d = 3000, 512
    a, b = planted(n, d, seed=77, noise=0.4)
    a[100:140] = a[99]                      # 40 exact duplicates of gallery row 99 ...
    b[100:140] = b[99]                      # ... and of query 99
    a[2000:2100] = a[1999] + (1e-7 * rng.standard_normal((100, d))).astype(np.float32)      # a dense cluster: fp64 decides, lists overflow
    b[1999:2100] = a[1999]
    for scale in (1.0, 25.0):
        sa, sb = (a * np.float32(scale)).astype(np.float32), (b * np.float32(scale)).astype(np.float32)
        got = ops.recall_bidir(torch.from_numpy(sa).cuda(), torch.from_numpy(sb).cuda(), [1, 5, 10]).cpu().numpy()
        assert np.array_equal(got, _hits_ref(sa, sb, [1, 5, 10])), (scale, got.tolist(), _hits_ref(sa, sb, [1, 5, 10]).tolist())
    g, q = SP.midpoint_case(d=512, depth=11)                  # gallery + one adversarial query whose true nearest row is gallery row 0
    n2 = 1024
    ga = np.concatenate([g, unit(rng.standard_normal((n2 - g.shape[0], 512))).astype(np.float32)]) if g.shape[0] < n2 else g[:n2]
    qb = ga.copy()
    qb[0] = q[0]                                              # pair 0 = the adversarial query against its true neighbour
    got = ops.recall_bidir(torch.from_numpy(ga).cuda(), torch.from_numpy(qb).cuda(), [1, 5, 10]).cpu().numpy()
    assert np.array_equal(got, _hits_ref(ga, qb, [1, 5, 10]))


@pytest.mark.parametrize("n", [10000, 50000])
def test_recall_bidir_rank_path_at_full_size_equals_the_two_step_form(n):
    """BASELINE sizes: the counters of the rank path == those of vtc_l2_topk_bidir + vtc_recall_hits_pair (whose ids are held to the fp64
    oracle elsewhere), through the drop-in metric and through sharded_recall(world = 1)."""
    from vtc_amd import dist as vdist
    from vtc_amd.host.metric import RecallAtK
    a, b = planted(n, 512, seed=n + 9, noise=9.0)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    m = RecallAtK("videos", "titles", [1, 5, 10])
    fast = m.compute_both(ta, tb)
    m2 = RecallAtK("videos", "titles", [1, 5, 10])
    m2.rank_path = False
    assert fast == m2.compute_both(ta, tb)
    r_ab, r_ba = vdist.sharded_recall(ta, tb, n, [1, 5, 10], 0, 1)
    assert r_ab == dict(fast[0]) and r_ba == dict(fast[1])
    print(f"[parity] rank path n={n}: R@1/5/10 {fast[0]} / {fast[1]}")
    assert 0.02 < fast[0][0][1] < 0.9999         # the case is neither trivial nor hopeless


def _play_rank_sharded(a, b, world, ks):
    """vtc_l2_recall_shard_rows -> exchange (by slicing, as vtc_amd/dist.py's all-to-all delivers it) -> vtc_l2_recall_shard_cols, the ranks
    played one after the other on this card; returns the summed counters [2, nk] (what the all-reduce leaves on every rank)."""
    from vtc_amd import dist as vdist
    from vtc_amd import ops
    n, d = a.shape
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    bounds = [vdist.shard_bounds(n, r, world) for r in range(world)]
    assert all(ops.recall_shard_supported(n, hi - lo, d) for lo, hi in bounds)
    rb = ops.sweep_row_block()
    nbp = -(-max(hi - lo for lo, hi in bounds) // rb)
    hits = torch.zeros(2, len(ks), dtype=torch.int64, device="cuda")
    planes = [ops.recall_shard_rows(ta, tb[lo:hi].contiguous(), lo, ks, nbp, hits[0]) for lo, hi in bounds]
    src_bounds = torch.tensor([lo for lo, _ in bounds] + [n], dtype=torch.int32, device="cuda")
    for lo, hi in bounds:
        recv = torch.stack([pl[:, :, lo:hi] for pl in planes]).contiguous()
        ops.recall_shard_cols(tb, ta[lo:hi].contiguous(), lo, ks, recv, src_bounds, hits[1])
    return hits.cpu().numpy()


@pytest.mark.parametrize("n,world,d,noise", [(4099, 3, 512, 0.6), (10000, 8, 512, 9.0), (6000, 2, 128, 0.3), (2050, 5, 64, 0.0)])
def test_rank_sharded_sweep_equals_the_fp64_oracle(n, world, d, noise):
    """The sharded sweep with the recall-only finish (round 5): per-rank partial counters of both directions, summed, == the fp64 oracle's and
    the single-GPU vtc_l2_recall_bidir's -- ragged shards (4099 = 1367 + 1366 + 1366: scalar plane loads, padded blocks whose keys are +inf,
    source bounds inside a block), 8 ranks at the 10k size, exact ties (noise 0: b == a) and several k sets."""
    from vtc_amd import ops
    a, b = planted(n, d, seed=n + world, noise=noise) if noise > 0 else (planted(n, d, seed=n)[0],) * 2
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    for ks in ([1, 5, 10], [2], [3, 7, 20, 50])[: 1 if n >= 5000 else 3]:
        got = _play_rank_sharded(a, b, world, ks)
        ref = _hits_ref(a, b, ks)
        assert np.array_equal(got, ref), (n, world, d, ks, got.tolist(), ref.tolist())
        one = ops.recall_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), ks).cpu().numpy()
        assert np.array_equal(got, one)


def test_rank_sharded_sweep_duplicates_clusters_and_scales():
    """Duplicates across shard borders (the tie-break compares GLOBAL indices), a dense near-duplicate cluster (lists overflow: brute force
    over the gathered set) and un-normalised rows, at 3 ragged ranks."""
    rng = np.random.default_rng(11)
    n, d, world = 3001, 512, 3
    a, b = planted(n, d, seed=78, noise=0.4)
    a[990:1040] = a[989]                    # shard border at 1001: duplicates on both sides of it
    b[990:1040] = b[989]
    a[2000:2100] = a[1999] + (1e-7 * rng.standard_normal((100, d))).astype(np.float32)
    b[1999:2100] = a[1999]
    for scale in (1.0, 25.0):
        sa, sb = (a * np.float32(scale)).astype(np.float32), (b * np.float32(scale)).astype(np.float32)
        got = _play_rank_sharded(sa, sb, world, [1, 5, 10])
        assert np.array_equal(got, _hits_ref(sa, sb, [1, 5, 10])), (scale, got.tolist(), _hits_ref(sa, sb, [1, 5, 10]).tolist())


def test_rank_sharded_sweep_at_50k_equals_the_single_gpu_counters():
    """BASELINE's 50k x 50k at 8 ranks of 6 250 rows: summed counters == vtc_l2_recall_bidir's (held to the two-step form and the oracle above)."""
    from vtc_amd import ops
    a, b = planted(50000, 512, seed=50017, noise=9.0)
    got = _play_rank_sharded(a, b, 8, [1, 5, 10])
    one = ops.recall_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), [1, 5, 10]).cpu().numpy()
    assert np.array_equal(got, one), (got.tolist(), one.tolist())
    assert 0.02 * 50000 < got[0][0] < 0.9999 * 50000


def test_rank_path_random_shapes_single_and_sharded():
    """Seeded random (n, d, k set, world): odd row counts (scalar plane loads), k up to 200 (beyond the 64 of the sorted-list form), duplicates,
    un-normalised rows -- vtc_l2_recall_bidir and the summed counters of the sharded form both equal the fp64 oracle's."""
    from vtc_amd import ops
    rng = np.random.default_rng(4242)
    for case in range(8):
        n = int(rng.integers(1024, 7000))
        d = int(rng.choice([64, 128, 512, 768]))
        world = int(rng.integers(2, 7))
        ks = sorted(int(k) for k in rng.choice([1, 2, 3, 5, 10, 17, 64, 65, 200], size=int(rng.integers(1, 5)), replace=False))
        a, b = planted(n, d, seed=3000 + case, noise=float(rng.choice([0.3, 0.8, 2.0, 6.0])))
        if case % 3 == 1:
            a[n // 3:n // 3 + 30] = a[5]                    # duplicates of an early row: ties broken by the lowest index
        if case % 3 == 2:
            sc = rng.uniform(0.5, 20.0, (n, 1)).astype(np.float32)
            a, b = (a * sc).astype(np.float32), (b * sc).astype(np.float32)
        ref = _hits_ref(a, b, ks)
        one = ops.recall_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), ks).cpu().numpy()
        assert np.array_equal(one, ref), (case, n, d, ks, one.tolist(), ref.tolist())
        got = _play_rank_sharded(a, b, world, ks)
        assert np.array_equal(got, ref), (case, n, d, world, ks, got.tolist(), ref.tolist())


def test_recall_entry_points_refuse_what_they_do_not_cover():
    """include/vtc_hip.h: n >= 1024, d % 64 == 0, 1 <= nk <= 4, 1 <= k <= n, rows inside the gathered set -- refused with a message, nothing launched."""
    import ctypes as C
    from vtc_amd import _lib as L
    lib = L.lib()
    assert lib.vtc_l2_recall_bidir_supported(1023, 512) == 0 and lib.vtc_l2_recall_bidir_supported(1024, 500) == 0
    assert lib.vtc_l2_recall_bidir_supported(1024, 512) == 1
    assert lib.vtc_l2_recall_shard_supported(4096, 512, 512) == 1 and lib.vtc_l2_recall_shard_supported(4096, 0, 512) == 0
    assert lib.vtc_l2_recall_shard_supported(1000, 500, 512) == 0 and lib.vtc_l2_recall_shard_supported(4096, 512, 96) == 0
    n, d = 2048, 64
    a = torch.randn(n, d, device="cuda")
    b = torch.randn(n, d, device="cuda")
    hits = torch.zeros(2, 4, dtype=torch.int64, device="cuda")
    ws = torch.empty(int(lib.vtc_l2_recall_bidir_workspace_bytes(n, d)), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream

    def call(ks, nk=None, ws_bytes=None):
        arr = (C.c_int * max(1, len(ks)))(*ks)
        return lib.vtc_l2_recall_bidir(a.data_ptr(), b.data_ptr(), n, d, arr, len(ks) if nk is None else nk, hits[0].data_ptr(), hits[1].data_ptr(),
                                       ws.data_ptr(), ws.numel() if ws_bytes is None else ws_bytes, s)
    assert call([1, 5, 10]) == 0
    for bad in (dict(ks=[0]), dict(ks=[n + 1]), dict(ks=[1, 2, 3, 4, 5]), dict(ks=[1], nk=0), dict(ks=[1], ws_bytes=1024)):
        assert call(**bad) != 0, bad
        assert lib.vtc_last_error(), bad
    torch.cuda.synchronize()
    planes = torch.empty(4, 2, n, dtype=torch.int32, device="cuda")
    ws2 = torch.empty(int(lib.vtc_l2_sweep_shard_workspace_bytes(n, 256, d)), dtype=torch.uint8, device="cuda")
    arr = (C.c_int * 1)(1)
    rc = lib.vtc_l2_recall_shard_rows(a.data_ptr(), b[:256].contiguous().data_ptr(), n, 256, n - 100, d, arr, 1, hits[0].data_ptr(), planes.data_ptr(), 2,
                                      ws2.data_ptr(), ws2.numel(), s)
    assert rc != 0 and b"outside" in lib.vtc_last_error()
    rc = lib.vtc_l2_recall_shard_rows(a.data_ptr(), b[:256].contiguous().data_ptr(), n, 256, 0, d, arr, 1, hits[0].data_ptr(), planes.data_ptr(), 1,
                                      ws2.data_ptr(), ws2.numel(), s)
    assert rc != 0 and b"nblk_pad" in lib.vtc_last_error()
    torch.cuda.synchronize()


def test_nonfinite_rows_are_misses_in_the_rank_path_and_rejected_by_the_metric():
    """a NaN target distance made every fp64 comparison of the recall-only rank path false -- rank 0, a hit at every k.
    (a) kernel level (ops.recall_bidir, no host check): a query / target pair with a non-finite embedding is a MISS at every k, and the
        other rows' counters are those of the oracle with NaN distances never closer than anything (what an exact search does with them);
        two planes (k <= 16) and four (k = 20), 1 100 and 4 099 rows;
    (b) host level: RecallAtK.compute / compute_both / result and dist.sharded_recall raise ValueError on non-finite features
        (check_finite, default on) instead of reporting a figure."""
    from vtc_amd import dist as vdist
    from vtc_amd import ops
    from vtc_amd.host.metric import RecallAtK
    for n, ks in ((1100, [1, 5, 10]), (4099, [1, 5, 10]), (1100, [1, 20])):
        rng = np.random.default_rng(n + len(ks))
        a = rng.standard_normal((n, 64)).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = (a + 0.9 * rng.standard_normal((n, 64))).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        a[5, 3] = np.nan
        b[n - 7, :] = np.inf
        raw = ops.recall_bidir(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), ks).cpu()
        got, marked = ops.split_recall_counters(raw)
        assert marked and int(raw[0, 0]) >> 40 and int(raw[1, 0]) >> 40          # VTC_RECALL_NONFINITE in both directions' first counter
        got = got.numpy()
        clean = ops.recall_bidir(torch.from_numpy(np.nan_to_num(a, nan=0.1)).cuda(), torch.from_numpy(np.nan_to_num(b, posinf=0.1)).cuda(), ks).cpu()
        assert not ops.split_recall_counters(clean)[1] and int(clean.max()) <= n
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        want = np.zeros((2, len(ks)), dtype=np.int64)
        for direction, (q, g) in enumerate(((b64, a64), (a64, b64))):
            for i0 in range(0, n, 512):
                with np.errstate(invalid="ignore", over="ignore"):
                    dd = ((q[i0:i0 + 512, None, :] - g[None, :, :]) ** 2).sum(-1)
                for r in range(dd.shape[0]):
                    i = i0 + r
                    dt = dd[r, i]
                    if not np.isfinite(dt):
                        continue                                       # a miss at every k
                    row = np.where(np.isfinite(dd[r]), dd[r], np.inf)
                    rank = int((row < dt).sum() + ((row == dt) & (np.arange(n) < i)).sum())
                    want[direction] += [rank < k for k in ks]
        np.testing.assert_array_equal(got, want)
    fa, fb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    m = RecallAtK("videos", "titles", [1, 5, 10])
    for call in (lambda: m.compute(fa, fb), lambda: m.compute_both(fa, fb), lambda: m.compute(a, b)):
        with pytest.raises(ValueError, match="non-finite values in the videos and titles features"):
            call()
    m.update(None, (fa, fb), None)
    with pytest.raises(ValueError, match="non-finite"):
        m.result()
    with pytest.raises(ValueError, match="non-finite"):
        vdist.sharded_recall(fa, fb, fa.shape[0], [1, 5, 10], 0, 1)
    ok = torch.from_numpy(np.nan_to_num(a, nan=0.0)).cuda()
    with pytest.raises(ValueError, match="in the titles features"):
        m.compute(ok, fb)
    # (c) the sharded kernels (three simulated ranks on this card): the marker rides in the partial counters of the rank that owns the bad
    # row -- what the all-reduce carries to every rank -- and the counters under it equal the single-GPU call's
    single, _ = ops.split_recall_counters(ops.recall_bidir(fa, fb, [1, 5, 10]).cpu())
    sharded_raw = torch.from_numpy(_play_rank_sharded(a, b, 3, [1, 5, 10]))
    sharded, marked = ops.split_recall_counters(sharded_raw)
    assert marked and torch.equal(sharded, single)
