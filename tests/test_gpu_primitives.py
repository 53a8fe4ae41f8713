"""GPU parity of the exported primitives (through the C ABI) against plain PyTorch fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _ops():
    from vtc_amd import _lib as L
    from vtc_amd import ops
    return L, ops


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 384, 192), (77, 512, 3072), (1000, 2304, 768), (5, 50, 64)])
def test_gemm_store_bias(dtype, M, N, K):
    L, ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    # integer-valued data first: exact in both dtypes, catches any fragment/lane-map mistake (asymmetric W)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N).float()[:, None] % 3
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = a.double() @ w.double().t() + bias.double()
    out = ops.gemm(a.cuda().to(dtype), w.cuda().to(dtype), bias.cuda(), out_dtype=torch.float32)
    assert torch.equal(out.cpu().double(), ref), (out.cpu().double() - ref).abs().max()
    # random data
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    ad, wd = a.cuda().to(dtype), w.cuda().to(dtype)
    ref = ad.float().cpu().double() @ wd.float().cpu().double().t() + bias.double()
    out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.float32).cpu().double()
    tol = 2e-5 if dtype == torch.float32 else 1e-4   # operands already rounded; fp32 accumulate
    assert (out - ref).abs().max() < tol * max(1.0, ref.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dtype):
    L, ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, N, K = 393 * 2, 256, 128
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    ad, wd = a.cuda().to(dtype), w.cuda().to(dtype)
    lin = ad.float().cpu().double() @ wd.float().cpu().double().t() + b.double()
    # QuickGELU, output in compute dtype
    out = ops.gemm(ad, wd, b.cuda(), epilogue=L.EPI_GELU).float().cpu().double()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (out - quick_gelu(lin)).abs().max() < tol
    # residual with cls-row skip (timesformer_clip_alt.py:149: temporal residual on patch tokens only)
    x0 = torch.randn(M, N, generator=g)
    x = x0.clone().cuda()
    ops.gemm(ad, wd, b.cuda(), epilogue=L.EPI_RESID, out=x, skip_mod=393)
    ref = x0.double() + lin
    ref[0::393] = x0[0::393].double()
    assert (x.cpu().double() - ref).abs().max() < 1e-4
    assert torch.equal(x.cpu()[0::393], x0[0::393])


@pytest.mark.parametrize("width", [128, 512, 768])
def test_layernorm(width):
    L, ops = _ops()
    g = torch.Generator().manual_seed(width)
    x = torch.randn(37, width, generator=g) * 3 + 1
    gm, bt = torch.randn(width, generator=g), torch.randn(width, generator=g)
    ref = torch.nn.functional.layer_norm(x, (width,), gm, bt, 1e-5)
    out = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda())
    assert (out.cpu() - ref).abs().max() < 5e-6
    outb = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), out_dtype=torch.bfloat16)
    assert (outb.float().cpu() - ref).abs().max() < 3e-2
    # row gather (EOT pooling / ln_post on cls rows)
    idx = torch.tensor([5, 0, 36, 5], dtype=torch.int32)
    outg = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), row_index=idx.cuda())
    assert (outg.cpu() - ref[idx.long()]).abs().max() < 5e-6
    outm = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), rows=4, row_mul=9)
    assert (outm.cpu() - ref[[0, 9, 18, 27]]).abs().max() < 5e-6


def ref_attention(q, k, v, causal):
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        Lq = q.shape[-2]
        s = s + torch.full((Lq, Lq), float("-inf")).triu_(1)
    return s.softmax(-1) @ v


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L_,causal", [(6, False), (8, False), (16, False), (24, True), (50, False), (77, True)])
def test_attention_contiguous(dtype, L_, causal):
    L, ops = _ops()
    heads, n_seq = 3, 5
    W = heads * 64
    g = torch.Generator().manual_seed(L_)
    qkv = torch.randn(n_seq * L_, 3 * W, generator=g)
    qd = qkv.cuda().to(dtype)
    q, k, v = qd.float().cpu().reshape(n_seq, L_, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ref_attention(q, k, v, causal).permute(0, 2, 1, 3).reshape(n_seq * L_, W)
    out = ops.attention(qd, n_seq, L_, heads, causal=causal).float().cpu()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (out - ref).abs().max() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L_,causal", [(81, False), (96, True), (129, True), (197, False), (257, False), (272, True)])
def test_attention_long_sequences_tiled(dtype, L_, causal):
    """80 < L <= 272 (ViT-B/16: 197, ViT-L/14: 257 tokens per frame): the K/V-tiled online-softmax kernel against
    softmax(q k^T / 8 [+ causal]) v in fp32 on the same (rounded) operands; sequence lengths on and off the 16- and 64-key grids."""
    L, ops = _ops()
    heads, n_seq = 2, 3
    W = heads * 64
    g = torch.Generator().manual_seed(L_)
    qkv = torch.randn(n_seq * L_, 3 * W, generator=g) * 1.5          # scores with a real spread: the running maximum moves
    qd = qkv.cuda().to(dtype)
    q, k, v = qd.float().cpu().reshape(n_seq, L_, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = ref_attention(q, k, v, causal).permute(0, 2, 1, 3).reshape(n_seq * L_, W)
    out = ops.attention(qd, n_seq, L_, heads, causal=causal).float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() < tol, float((out - ref).abs().max())


def test_attention_rejects_sequences_beyond_the_tiled_kernel():
    L, ops = _ops()
    qd = torch.zeros(273, 3 * 64, device="cuda")
    with pytest.raises(RuntimeError, match="272"):
        ops.attention(qd, 1, 273, 1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_single_query_attention_vs_fp32_reference(dtype):
    """vtc_single_query_attention (the last block's attention): one projected query per sequence over the keys / values of
    the packed qkv buffer -- contiguous sequences (image tower), the space branch's [cls, frame patches] map with the query shared by
    an item's F sequences, and the text tower's EOT query over rows base .. eot (ragged offsets and dense) -- against
    softmax(q k^T / 8) v in fp32 on the same (rounded) operands.  The Q third of the buffer is poisoned: it must not be read."""
    L, ops = _ops()
    heads = 3
    W = heads * 64
    tol = 2e-5
    g = torch.Generator().manual_seed(3)

    def ref(qrow, krows, vrows):             # [W], [n, W], [n, W] -> [W]
        out = []
        for h in range(heads):
            sl = slice(64 * h, 64 * h + 64)
            p = ((krows[:, sl] @ qrow[sl]) * 0.125).softmax(0)
            out.append(p @ vrows[:, sl])
        return torch.cat(out)

    # contiguous sequences of L tokens, query = row o of q
    for L_ in (1, 7, 50, 77):
        n = 5
        qkv = torch.randn(n * L_, 3 * W, generator=g)
        qkv[:, :W] = float("nan")
        q = torch.randn(n, W, generator=g)
        qd, kd = q.cuda().to(dtype), qkv.cuda().to(dtype)
        got = ops.single_query_attention(kd, qd, n, L_, heads).cpu()
        kf = kd.float().cpu()
        want = torch.stack([ref(qd.float().cpu()[o], kf[o * L_:(o + 1) * L_, W:2 * W], kf[o * L_:(o + 1) * L_, 2 * W:]) for o in range(n)])
        assert torch.isfinite(got).all() and (got - want).abs().max() < tol, (L_, (got - want).abs().max())
    # space branch: item b, frame t: keys = cls row b T, then rows b T + 1 + n F + t; the query of (b, t) is q[b]
    B, P, F = 3, 4, 8
    T = 1 + P * F
    qkv = torch.randn(B * T, 3 * W, generator=g)
    qkv[:, :W] = float("nan")
    q = torch.randn(B, W, generator=g)
    qd, kd = q.cuda().to(dtype), qkv.cuda().to(dtype)
    got = ops.single_query_attention(kd, qd, B * F, 1 + P, heads, s2=F, a0=0, a1=T, a2=0, a3=1, pstride=F).cpu()
    kf = kd.float().cpu()
    for b in range(B):
        for t in range(F):
            rows = [b * T] + [b * T + 1 + n_ * F + t for n_ in range(P)]
            want = ref(qd.float().cpu()[b], kf[rows, W:2 * W], kf[rows, 2 * W:])
            assert (got[b * F + t] - want).abs().max() < tol
    # text: sequence o = rows base .. eot[o] (ragged: base = offs[o]; dense: base = o ctx)
    ctx = 24
    lens = torch.tensor([1, 24, 5, 17], dtype=torch.int32)
    offs = torch.zeros(5, dtype=torch.int32)
    offs[1:] = torch.cumsum(lens, 0)
    for dense in (False, True):
        rows_total = 4 * ctx if dense else int(offs[-1])
        qkv = torch.randn(rows_total, 3 * W, generator=g)
        qkv[:, :W] = float("nan")
        q = torch.randn(4, W, generator=g)
        base = torch.arange(4, dtype=torch.int32) * ctx if dense else offs[:4]
        eot = (base + lens - 1).to(torch.int32)
        qd, kd = q.cuda().to(dtype), qkv.cuda().to(dtype)
        got = ops.single_query_attention(kd, qd, 4, 0, heads, eot=eot.cuda(), offs=None if dense else offs.cuda(), ctx=ctx).cpu()
        kf = kd.float().cpu()
        for o in range(4):
            r = slice(int(base[o]), int(eot[o]) + 1)
            want = ref(qd.float().cpu()[o], kf[r, W:2 * W], kf[r, 2 * W:])
            assert (got[o] - want).abs().max() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("F", [1, 2, 8, 16])
def test_attention_time_and_space_row_maps(dtype, F):
    """Token order of model/timesformer_clip_alt.py:271-275: row b*T = cls, row b*T + 1 + n*F + t."""
    L, ops = _ops()
    heads, B, P = 2, 3, 4
    W, T = heads * 64, 1 + P * F
    g = torch.Generator().manual_seed(F)
    qkv = torch.randn(B * T, 3 * W, generator=g)
    qd = qkv.cuda().to(dtype)
    x = qd.float().cpu().reshape(B, T, 3, heads, 64)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    # time: sequences (b, n), tokens t
    pt = x[:, 1:].reshape(B, P, F, 3, heads, 64)
    q, k, v = pt.permute(3, 0, 1, 4, 2, 5)                       # [3, B, P, heads, F, 64]
    ref_t = ref_attention(q, k, v, False).permute(0, 1, 3, 2, 4).reshape(B, P * F, W)
    out = ops.attention(qd, B * P, F, heads, s2=P, a0=1, a1=T, a2=F, a3=0, pstride=1).float().cpu().reshape(B, T, W)
    assert (out[:, 1:] - ref_t).abs().max() < tol
    assert out[:, 0].abs().max() == 0                            # cls rows untouched by the temporal branch
    # space: sequences (b, t): [cls, patches of frame t]; cls outputs go to cls_out[(b t)]
    ps = pt.permute(0, 2, 1, 3, 4, 5)                            # [B, F, P, 3, heads, 64]
    cls = x[:, 0:1].unsqueeze(1).expand(B, F, 1, 3, heads, 64)
    seq = torch.cat([cls, ps], dim=2)                            # [B, F, 1+P, 3, heads, 64]
    q, k, v = seq.permute(3, 0, 1, 4, 2, 5)
    ref_s = ref_attention(q, k, v, False).permute(0, 1, 3, 2, 4).reshape(B, F, 1 + P, W)
    cls_out = torch.zeros(B * F, W, device="cuda")
    out = ops.attention(qd, B * F, 1 + P, heads, s2=F, a0=0, a1=T, a2=0, a3=1, pstride=F, cls_out=cls_out)
    out = out.float().cpu().reshape(B, T, W)
    got = out[:, 1:].reshape(B, P, F, W).permute(0, 2, 1, 3)
    assert (got - ref_s[:, :, 1:]).abs().max() < tol
    assert (cls_out.cpu().reshape(B, F, W) - ref_s[:, :, 0]).abs().max() < tol


# ---- the phased 256x256 bf16 kernel (chosen when the problem has >= 384 tiles of 256x256) -------------------
@pytest.mark.parametrize("M,N,K", [(8192, 3072, 512), (8200, 3000, 768), (12800, 2304, 64), (16384, 1536, 1024)])
def test_gemm_phased_store_bias(M, N, K):
    """Interior and edge tiles (M, N not multiples of 256), one K-tile and many, fp32 and bf16 outputs."""
    L, ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    # integer-valued: exact, catches any quarter / fragment / lane-map mistake; W rows and A rows all distinct
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + (torch.arange(N).float()[:, None] % 3)
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = (a @ w.t() + bias)                      # |values| < 2^24: exact in fp32
    out = ops.gemm(a.cuda().bfloat16(), w.cuda().bfloat16(), bias.cuda(), out_dtype=torch.float32).cpu()
    assert torch.equal(out, ref), (out - ref).abs().max()
    # random data, bf16 output
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    ad, wd = a.cuda().bfloat16(), w.cuda().bfloat16()
    ref = ad.float() @ wd.float().t() + bias.cuda()
    out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.bfloat16).float()
    assert (out - ref).abs().max() < 2e-2 * max(1.0, float(ref.abs().max()))
    out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.float32)
    assert (out - ref).abs().max() < 2e-4 * max(1.0, float(ref.abs().max()))


def test_gemm_phased_epilogues():
    L, ops = _ops()
    g = torch.Generator().manual_seed(11)
    M, N, K = 393 * 24, 3072, 256          # 37 x 12 tiles, last M tile partial
    a, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    ad, wd = a.cuda().bfloat16(), w.cuda().bfloat16()
    lin = ad.float() @ wd.float().t() + b.cuda()
    out = ops.gemm(ad, wd, b.cuda(), epilogue=L.EPI_GELU).float()
    assert (out - quick_gelu(lin)).abs().max() < 2e-2
    x0 = torch.randn(M, N, generator=g).cuda()
    x = x0.clone()
    ops.gemm(ad, wd, b.cuda(), epilogue=L.EPI_RESID, out=x, skip_mod=393)
    ref = x0 + lin
    ref[0::393] = x0[0::393]
    assert (x - ref).abs().max() < 2e-4
    assert torch.equal(x[0::393], x0[0::393])
    # many tiles per workgroup (persistent walk + cross-tile prefetch), result independent of the walk
    M2 = 256 * 40
    a2 = torch.randint(-2, 3, (M2, 128), generator=g).float()
    w2 = torch.randint(-2, 3, (4096, 128), generator=g).float()
    out = ops.gemm(a2.cuda().bfloat16(), w2.cuda().bfloat16(), None, out_dtype=torch.float32).cpu()
    assert torch.equal(out, a2 @ w2.t())


def test_gemm_random_shapes_all_kernels():
    """Randomised shapes over every kernel configuration (64x64, 128x128, 256x256 phased; forced through
    VTC_GEMM_TILE-independent heuristics by size), ragged edges in M and N, K from one K-tile up, every epilogue --
    integer-valued operands make the fp32 result exact, so any indexing mistake shows as a mismatch."""
    L, ops = _ops()
    rng = np.random.default_rng(2024)
    cases = []
    for _ in range(10):      # small / medium: 64x64 and 128x128 kernels
        cases.append((int(rng.integers(1, 700)), int(rng.integers(1, 700)), 64 * int(rng.integers(1, 6))))
    for _ in range(6):       # large: phased kernel, ragged edges
        cases.append((int(rng.integers(6000, 9000)), int(rng.integers(2500, 3400)), 64 * int(rng.integers(1, 5))))
    for (M, N, K) in cases:
        g = torch.Generator().manual_seed(M * 31 + N)
        a = torch.randint(-2, 3, (M, K), generator=g).float()
        w = torch.randint(-2, 3, (N, K), generator=g).float()
        bias = torch.randint(-4, 5, (N,), generator=g).float()
        ad, wd = a.cuda().bfloat16(), w.cuda().bfloat16()
        ref = (a.cuda() @ w.cuda().t()) + bias.cuda()            # exact in fp32 (|values| small integers)
        out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.float32)
        assert torch.equal(out, ref), (M, N, K, "store f32", float((out - ref).abs().max()))
        out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.bfloat16)
        assert torch.equal(out.float(), ref.bfloat16().float()), (M, N, K, "store bf16")
        x0 = torch.randint(-8, 9, (M, N), generator=g).float().cuda()
        x = x0.clone()
        ops.gemm(ad, wd, bias.cuda(), epilogue=L.EPI_RESID, out=x, skip_mod=7)
        want = x0 + ref
        want[0::7] = x0[0::7]
        assert torch.equal(x, want), (M, N, K, "resid")
        out = ops.gemm(ad, wd, bias.cuda(), epilogue=L.EPI_GELU, out_dtype=torch.float32)
        assert (out - quick_gelu(ref)).abs().max() < 2e-2 * max(1.0, float(ref.abs().max())), (M, N, K, "gelu")


# ---- IEEE-half operands (VTC_F16): the text tower's blocks in bf16 mode --------------------------------------------
@pytest.mark.parametrize("M,N,K", [(300, 384, 192), (77, 512, 3072), (8200, 3000, 768)])
def test_gemm_f16_operands(M, N, K):
    """Same fragment maps as bf16 (integer-exact check, asymmetric W, small kernels and the phased 256x256 one), every
    epilogue; 16-bit outputs come out in IEEE half."""
    L, ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float() + torch.arange(N).float()[:, None] % 3
    bias = torch.randint(-5, 6, (N,), generator=g).float()
    ref = (a.cuda() @ w.cuda().t()) + bias.cuda()
    ad, wd = a.cuda().half(), w.cuda().half()
    out = ops.gemm(ad, wd, bias.cuda(), out_dtype=torch.float32)
    assert torch.equal(out, ref), float((out - ref).abs().max())
    out16 = ops.gemm(ad, wd, bias.cuda())
    assert out16.dtype == torch.float16 and torch.equal(out16.float(), ref.half().float())
    x0 = torch.randint(-8, 9, (M, N), generator=g).float().cuda()
    x = x0.clone()
    ops.gemm(ad, wd, bias.cuda(), epilogue=L.EPI_RESID, out=x, skip_mod=7)
    want = x0 + ref
    want[0::7] = x0[0::7]
    assert torch.equal(x, want)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5
    ad, wd = a.cuda().half(), w.cuda().half()
    lin = ad.float() @ wd.float().t() + bias.cuda()
    out = ops.gemm(ad, wd, bias.cuda(), epilogue=L.EPI_GELU).float()
    assert (out - quick_gelu(lin)).abs().max() < 4e-3 * max(1.0, float(lin.abs().max()))     # half: 2^-11 relative


def test_layernorm_and_attention_f16():
    L, ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 512, generator=g) * 3 + 1
    gm, bt = torch.randn(512, generator=g), torch.randn(512, generator=g)
    ref = torch.nn.functional.layer_norm(x, (512,), gm, bt, 1e-5)
    out = ops.layernorm(x.cuda(), gm.cuda(), bt.cuda(), out_dtype=torch.float16)
    assert out.dtype == torch.float16 and (out.float().cpu() - ref).abs().max() < 6e-3
    assert torch.equal(out.cpu(), ops.layernorm(x.cuda(), gm.cuda(), bt.cuda()).half().cpu())   # = round-to-nearest of the fp32 result
    for L_, causal in ((8, False), (50, False), (77, True)):
        heads, n_seq = 3, 4
        W = heads * 64
        qkv = torch.randn(n_seq * L_, 3 * W, generator=g).cuda().half()
        q, k, v = qkv.float().cpu().reshape(n_seq, L_, 3, heads, 64).permute(2, 0, 3, 1, 4)
        ref = ref_attention(q, k, v, causal).permute(0, 2, 1, 3).reshape(n_seq * L_, W)
        out = ops.attention(qkv, n_seq, L_, heads, causal=causal)
        assert out.dtype == torch.float16 and (out.float().cpu() - ref).abs().max() < 3e-3


def test_entry_points_from_two_threads_and_two_streams():
    """SURVEY 8b "re-entrant and per-device/stream safe": nn.DataParallel calls forward from one thread per replica
    (train.py:77-80).  Two host threads, each on its own HIP stream, push GEMMs (both kernel families, dynamic-LDS
    attribute set on first use), LayerNorms and attentions through the C ABI concurrently; every result must equal the
    single-threaded one bit for bit."""
    import threading
    L, ops = _ops()
    g = torch.Generator().manual_seed(17)
    jobs = []
    for (M, N, K) in ((4096, 3072, 512), (300, 384, 192), (9000, 2560, 256)):
        a = torch.randint(-3, 4, (M, K), generator=g).float().cuda().bfloat16()
        w = torch.randint(-3, 4, (N, K), generator=g).float().cuda().bfloat16()
        jobs.append((a, w))
    qkv = torch.randn(6 * 50, 3 * 128, generator=g).cuda().bfloat16()
    x = torch.randn(999, 768, generator=g).cuda()
    gm, bt = torch.randn(768, generator=g).cuda(), torch.randn(768, generator=g).cuda()

    def work():
        outs = [ops.gemm(a, w, None, out_dtype=torch.float32) for a, w in jobs]
        outs.append(ops.attention(qkv, 6, 50, 2).float())
        outs.append(ops.layernorm(x, gm, bt))
        return outs

    want = work()
    torch.cuda.synchronize()
    results, errors = {}, []

    def runner(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(8):
                    results[i] = work()
            st.synchronize()
        except Exception as e:     # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=runner, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for i in range(2):
        for o, wv in zip(results[i], want):
            assert torch.equal(o, wv)


# ---- residual GEMM + the following LayerNorm in one launch (EPI_RESID_LN) ------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,skip", [(100608, 768, 768, 393), (100608, 768, 3072, 0), (118272, 512, 512, 0), (56789, 512, 2048, 0),
                                        (70001, 1024, 256, 7)])
def test_gemm_resid_layernorm_equals_two_launches(dtype, M, N, K, skip):
    """vtc_gemm_resid_layernorm == vtc_gemm(RESID) then vtc_layernorm, BIT FOR BIT, on the tower shapes (W = 768: three
    column tiles per row block; W = 512: two; 1024: four), ragged M, the cls-row skip of the temporal branch.  The LayerNorm is
    done by whichever column tile of a 256-row block finishes last, reading the block back through write-through stores and
    L1-bypassing loads: run several times (different arrival orders) and with the consumer's caches warm (the residual rows
    are read by plain loads just before: Guideline 16, Pitfall 3)."""
    L, ops = _ops()
    g = torch.Generator().manual_seed(M % 1000 + N)
    a = torch.randn(M, K, generator=g).cuda().to(dtype)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).cuda().to(dtype)
    b = torch.randn(N, generator=g).cuda()
    ln_g, ln_b = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    x0 = (torch.randn(M, N, generator=g) * 2).cuda()
    x_ref = x0.clone()
    ops.gemm(a, w, b, epilogue=L.EPI_RESID, out=x_ref, skip_mod=skip)
    h_ref = ops.layernorm(x_ref, ln_g, ln_b, out_dtype=dtype)
    for rep in range(4):
        x = x0.clone()
        warm = float(x.sum())                                  # plain loads of every line first
        h = ops.gemm_resid_layernorm(a, w, b, x, ln_g, ln_b, skip_mod=skip)
        assert torch.equal(x, x_ref), (rep, float((x - x_ref).abs().max()))
        assert torch.equal(h, h_ref), (rep, int((h != h_ref).sum()), float((h.float() - h_ref.float()).abs().max()))
    # in place: ln_out aliases the A operand (what the towers do)
    x = x0.clone()
    a2 = a.clone() if K == N else None
    if a2 is not None:
        ops.gemm_resid_layernorm(a2, w, b, x, ln_g, ln_b, skip_mod=skip, ln_out=a2)
        assert torch.equal(x, x_ref) and torch.equal(a2, h_ref)


@pytest.mark.parametrize("hd,L_", [(96, 6), (32, 16), (128, 3), (64, 6)])
def test_cam_attention_any_head_dim_vs_fp32_reference(hd, L_):
    """Round 4: the Context Adapter Module with head_dim != 64 (ViT-L/14's 768-d features at the reference's default n_heads = 8:
    head_dim 96) runs its 1 + nc tokens through a generic short-sequence core; checked through vtc_cam_forward against the
    oracle's CAM (model/model.py:141-205 restated) on random CAM weights."""
    from oracle import arch as A
    from oracle import model_ref as M
    from vtc_amd import towers
    heads = 4
    D = heads * hd
    nc = L_ - 1
    a = A.ClipArch(embed_dim=D, transformer_width=D)
    sd = A.synth_cam(a, 5, n_layers=2)
    g = torch.Generator().manual_seed(hd + L_)
    for k in list(sd):
        if k.endswith("out_proj.weight") or k.endswith("c_proj.weight"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    B = 5
    main = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=-1)
    aux = torch.randn(nc, B, D, generator=g)
    want = M.adapt_feature(main, aux, sd, n_heads=heads)
    pc = towers.PackedCam({k: v.cuda() for k, v in sd.items()}, torch.float32, heads, True, None)
    comments = torch.full((B, nc, 77), 5, dtype=torch.int64)      # no empty comment: aux is used as given
    comments[..., 0] = 49406
    got = pc.forward(main.cuda(), aux.permute(1, 0, 2).reshape(B * nc, D).contiguous().cuda(), comments.cuda()).cpu()
    assert (got - want).abs().max() < 1e-5, float((got - want).abs().max())
