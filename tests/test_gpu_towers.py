"""GPU parity of the towers, the CAM and the four drop-in wrappers.

Every case runs the HIP path through the C ABI and compares with (a) the committed golden
vectors produced by the reference's own Python and (b) the oracle run live on the CPU.
Tolerances are BASELINE.json's: 1e-5 (fp32) / 1e-3 (bf16) on unit-norm embeddings and on the
cosine similarity (sim / exp(logit_scale))."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import arch as A
from oracle import clip_ref as CR
from oracle import model_ref as M
from oracle import timesformer_ref as T

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
EXTENDED = __import__("os").environ.get("VTC_TEST_EXTENDED") == "1" or "--extended" in __import__("sys").argv
ARCH = {"TINY": A.TINY, "VIT_B32": A.VIT_B32, "VIT_B16": A.VIT_B16, "VIT_L14": A.VIT_L14}
DTYPES = [torch.float32, torch.bfloat16]
DTYPES_H = DTYPES + [torch.float16]      # + the IEEE-half mode of the vision towers (round 6): the tower goldens and the larger model types


def tol_for(dtype, embed_dim=512):
    """BASELINE.json: 1e-5 (fp32) / 1e-3 (bf16) on the unit-norm 512-d embeddings of the real model,
    i.e. 2.3 % of the RMS element 1/sqrt(512) in bf16.  The TINY oracle architecture has 128-d
    embeddings whose elements are 2x larger, so the same relative bf16 accuracy is 2e-3 there."""
    if dtype == torch.float32:
        return 1e-5
    return 1e-3 * (512 / embed_dim) ** 0.5


def cuda_sd(sd):
    return {k: v.cuda() for k, v in sd.items()}


def unit(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def report(name, err, tol):
    print(f"[parity] {name}: max abs err {err:.3e} (tol {tol:.0e})")
    assert err < tol, f"{name}: {err} >= {tol}"


def report_l2(name, got, want, dtype):
    """A bound that does not scale with the embedding width (`tol_for` relaxes the max-abs bound to 2e-3 on the
    128-d TINY architecture): the L2 norm of the error of a UNIT-NORM embedding.  1e-3 max abs on 512 dimensions allows 2.3e-2; the
    16-bit paths sit at 3-8e-3 on either architecture, fp32 at 1e-6: asserted at 1e-2 / 1e-5 whatever the width."""
    e = float(np.linalg.norm(unit(np.asarray(got, dtype=np.float64)) - unit(np.asarray(want, dtype=np.float64)), axis=-1).max())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    print(f"[parity] {name}: L2 error of the unit-norm embedding {e:.3e} (tol {tol:.0e}, width-independent)")
    assert e < tol, f"{name}: L2 {e} >= {tol}"


def report_text(name, got, want, dtype, embed_dim):
    """Text-tower features: BASELINE.json's tolerance, 1e-5 (fp32) / 1e-3 (bf16 mode) max abs on the unit-norm
    embedding (x sqrt(512 / D) for the 128-d TINY architecture).  An all-bf16 text tower sits at rms 3.2e-4 / max
    1.1-1.4e-3 (tests/bf16_floor_study.py: the operand-rounding floor, no kernel involved), so in bf16 mode the text
    blocks run with IEEE-half operands (vtc_amd.towers.TEXT_HALF_LAYERS) -- same rate, 3 more significant bits."""
    d = np.abs(got - want)
    tol = tol_for(dtype, embed_dim)
    rms = float(np.sqrt((d ** 2).mean()))
    print(f"[parity] {name}: max abs err {d.max():.3e} rms {rms:.3e} (tol {tol:.1e})")
    assert d.max() < tol, f"{name}: max {d.max()} rms {rms} >= {tol}"


@pytest.mark.parametrize("dtype", DTYPES_H)
@pytest.mark.parametrize("fname", golden_files("tower_alt_"))
def test_timesformer_tower_vs_golden(fname, dtype):
    from vtc_amd import towers
    case, g = load_golden(fname)
    a = ARCH[case["arch"]]
    sd = A.synth_visual(a, case["wseed"], nframes=case["nframes"], prefix="v.")
    x = A.synth_pixels((case["B"], case["nframes"], 3, a.image_resolution, a.image_resolution), case["xseed"])
    for fuse in (False, True):
        pv = towers.PackedVision(cuda_sd(sd), "v.", dtype, fuse_temporal=fuse)
        out = pv.forward(x.cuda()).cpu().numpy()
        # compare as the wrappers consume it: L2-normalised embedding (model.py:501)
        report(f"{fname} {dtype} fuse={fuse}", np.abs(unit(out) - unit(g["out"])).max(),
                   tol_for(dtype, a.embed_dim))      # fp32: 1e-5 with the pre-multiplied temporal_fc o out_proj too
        report_l2(f"{fname} {dtype} fuse={fuse}", out, g["out"], dtype)
        if dtype == torch.float32 and not fuse:
            assert np.abs(out - g["out"]).max() < 2e-5 * max(1.0, np.abs(g["out"]).max())


@pytest.mark.parametrize("dtype", DTYPES_H)
@pytest.mark.parametrize("fname", golden_files("tower_v1_"))
def test_timesformer_v1_tower_vs_golden(fname, dtype):
    """model/timesformer_clip.py (older variant: global cls attention, no temporal_fc)."""
    from vtc_amd import towers
    case, g = load_golden(fname)
    a = ARCH[case["arch"]]
    sd = A.synth_visual(a, case["wseed"], nframes=case["nframes"], prefix="v.", variant="v1")
    x = A.synth_pixels((case["B"], case["nframes"], 3, a.image_resolution, a.image_resolution), case["xseed"])
    pv = towers.PackedVision(cuda_sd(sd), "v.", dtype)
    assert pv.w.variant == 1
    out = pv.forward(x.cuda()).cpu().numpy()
    report(f"{fname} {dtype}", np.abs(unit(out) - unit(g["out"])).max(), tol_for(dtype, a.embed_dim))
    report_l2(f"{fname} {dtype}", out, g["out"], dtype)
    if dtype == torch.float32:
        assert np.abs(out - g["out"]).max() < 2e-5 * max(1.0, np.abs(g["out"]).max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_vit_and_text_towers_vs_oracle(dtype):
    from vtc_amd import towers
    for a, B, S in ((A.TINY, 5, 9), (A.VIT_B32, 3, 7)):
        sd = {}
        sd.update(A.synth_visual(a, 51, prefix="model.visual."))
        sd.update(A.synth_text(a, 52, prefix="model."))
        img = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), 53)
        txt = A.synth_tokens(S, a, 54, empty_frac=0.25)
        ref_v = CR.encode_image(img, sd, a, "model.visual.").numpy()
        ref_t = CR.encode_text(txt, sd, a, "model.").numpy()
        pv = towers.PackedVision(cuda_sd(sd), "model.visual.", dtype)
        pt = towers.PackedText(cuda_sd(sd), "model.", dtype, heads=a.transformer_heads)
        out_v = pv.forward(img.cuda()).cpu().numpy()
        out_t = pt.forward(txt.cuda()).cpu().numpy()
        report(f"ViT {a.vision_width} {dtype}", np.abs(unit(out_v) - unit(ref_v)).max(), tol_for(dtype, a.embed_dim))
        report_l2(f"ViT {a.vision_width} {dtype}", out_v, ref_v, dtype)
        report_text(f"text {a.transformer_width} {dtype}", unit(out_t), unit(ref_t), dtype, a.embed_dim)
        if dtype == torch.bfloat16:  # bf16 pixel input (BASELINE: pixels cast to bf16 for bf16 runs)
            out_vb = pv.forward(img.cuda().bfloat16()).cpu().numpy()
            report(f"ViT bf16-pixels {a.vision_width}", np.abs(unit(out_vb) - unit(ref_v)).max(), 2 * tol_for(dtype, a.embed_dim))


@pytest.mark.parametrize("dtype", DTYPES)
def test_ragged_text_tower_equals_dense(dtype):
    """Tokens after EOT cannot reach the EOT feature through a causal tower: the ragged path (only tokens
    0..EOT computed) must reproduce the dense path and the oracle, including 2-token empty strings, full-length
    rows and rows without any EOT (argmax = position of the largest id)."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_text(a, 52, prefix="model.")
    txt = A.synth_tokens(37, a, 54, empty_frac=0.25)
    txt[3, 1:76] = torch.randint(1, A.SOT, (75,)); txt[3, 76] = A.EOT          # full length
    txt[5] = torch.randint(1, 1000, (77,)); txt[5, 40] = 48000                 # no EOT: argmax picks position 40
    pt = towers.PackedText(cuda_sd(sd), "model.", dtype, heads=a.transformer_heads)
    dense = pt.forward(txt.cuda(), ragged=False).cpu().numpy()
    ragged = pt.forward(txt.cuda(), ragged=True).cpu().numpy()
    ref = CR.encode_text(txt, sd, a, "model.").numpy()
    if dtype == torch.float32:
        assert np.abs(ragged - dense).max() < 2e-5 * np.abs(dense).max()
        assert np.abs(unit(ragged) - unit(ref)).max() < 1e-5
    else:
        report_text("ragged text bf16", unit(ragged), unit(ref), dtype, a.embed_dim)
        assert np.abs(unit(ragged) - unit(dense)).max() < 1e-3


def test_identity_at_init_timesformer_equals_vit_gpu():
    """SURVEY 4 known answer: temporal_fc = 0, temporal_embed = 0, identical frames => TimeSformer == ViT."""
    from vtc_amd import towers
    a = A.TINY
    sd = A.synth_visual(a, 5, nframes=4)
    for k in list(sd):
        if "temporal_fc" in k or k == "temporal_embed":
            sd[k] = torch.zeros_like(sd[k])
    img = A.synth_pixels((2, 1, 3, a.image_resolution, a.image_resolution), 6)
    vid = img.expand(2, 4, 3, a.image_resolution, a.image_resolution).contiguous()
    sd_vit = {k: v for k, v in sd.items() if "time" not in k and "temporal" not in k}
    tf = towers.PackedVision(cuda_sd({"v." + k: v for k, v in sd.items()}), "v.", torch.float32).forward(vid.cuda()).cpu()
    vit = towers.PackedVision(cuda_sd({"v." + k: v for k, v in sd_vit.items()}), "v.", torch.float32).forward(img[:, 0].cuda()).cpu()
    assert (tf - vit).abs().max() < 1e-5


def build_wrapper(case, dtype):
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    a = ARCH[case["arch"]]
    cls = {"clip": HM.PretrainedCLIP, "clip_finaltf": HM.PretrainedCLIP_finaltf,
           "timesformer": HM.PretrainedCLIP_TimeSformer, "timesformer_finaltf": HM.PretrainedCLIP_TimeSformer_finaltf}[case["model"]]
    m = cls(model_type=ClipConfig(**asdict(a)), **case["ctor"])
    bn_stats = case["ctor"].get("residual_activation") in ("sub_mean", "bn")
    m.load_state_dict(A.synth_model(a, case["wseed"], case["model"], nframes=8, bn_stats=bn_stats), strict=True)   # eval.py:90-91
    m = m.eval().cuda()
    m.compute_dtype = dtype
    return m, a


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fname", golden_files("wrap_"))
def test_wrappers_vs_golden(fname, dtype):
    case, g = load_golden(fname)
    m, a = build_wrapper(case, dtype)
    B = case["B"]
    vis = A.synth_pixels(case["vis_shape"], case["xseed"]).cuda()
    title = A.synth_tokens(B, a, case["tseed"]).cuda()
    comments = A.synth_tokens(B * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(B, 5, -1).cuda()
    out = m(vis, title, comments) if case["comments"] else m(vis, title)
    fv, ft, sim = (o.cpu().numpy() for o in out)
    tol = tol_for(dtype, a.embed_dim)
    report(f"{fname} feats_vis {dtype}", np.abs(fv - g["feats_vis"]).max(), tol)
    if dtype == torch.float32:
        report(f"{fname} feats_text {dtype}", np.abs(ft - g["feats_text"]).max(), tol)
    else:
        report_text(f"{fname} feats_text {dtype}", ft, g["feats_text"], dtype, a.embed_dim)
    scale = float(np.exp(np.log(1 / 0.07)))
    report(f"{fname} cos-sim {dtype}", np.abs(sim - g["sim"]).max() / scale, tol)
    np.testing.assert_allclose(np.linalg.norm(fv, axis=-1), 1.0, atol=1e-5)
    # R@1 ranks of the batch similarity identical to the reference's
    if np.sort(g["sim"], axis=1)[:, -1].min() - np.sort(g["sim"], axis=1)[:, -2].max() > 0.1:
        assert np.array_equal(sim.argmax(1), g["sim"].argmax(1))


def test_cam_at_init_and_branch_isolation():
    """tests/test_pretrained_clip.py:36-42,74-85 of the reference, on the HIP path:
    skip == plain CLIP; only the adapted modality changes; image feature independent of the title."""
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    case, g = load_golden("cam_at_init_tiny.npz")
    a = A.TINY
    cfg = ClipConfig(**asdict(a))
    sd = A.synth_model(a, case["wseed"], "clip_finaltf", cam_at_init=True)
    B = case["B"]
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), case["xseed"]).cuda()
    title = A.synth_tokens(B, a, case["tseed"]).cuda()
    title2 = A.synth_tokens(B, a, case["tseed"] + 100).cuda()
    comments = A.synth_tokens(B * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(B, 5, -1).cuda()
    outs = {}
    for br in ("skip", "image", "text"):
        m = HM.PretrainedCLIP_finaltf(model_type=cfg, branch_to_adapt_val=br, n_heads=2)
        m.load_state_dict(sd, strict=True)
        m = m.eval().cuda()
        m.compute_dtype = torch.float32
        outs[br] = tuple(o.cpu() for o in m(vis, title, comments))
        if br == "image":
            imv2, titlev2, _ = (o.cpu() for o in m(vis, title2, comments))
    plain = HM.PretrainedCLIP(model_type=cfg)
    plain.load_state_dict({k: v for k, v in sd.items() if k.startswith("model.")}, strict=True)
    plain = plain.eval().cuda()
    plain.compute_dtype = torch.float32
    pim, ptx, _ = (o.cpu() for o in plain(vis, title))
    assert torch.allclose(outs["skip"][0], pim, atol=1e-6) and torch.allclose(outs["skip"][1], ptx, atol=1e-6)
    assert torch.allclose(outs["skip"][0], outs["text"][0], atol=1e-6)       # image unchanged when adapting text
    assert torch.allclose(outs["skip"][1], outs["image"][1], atol=1e-6)
    assert not torch.allclose(outs["image"][0], outs["skip"][0], atol=1e-4)
    assert not torch.allclose(outs["text"][1], outs["skip"][1], atol=1e-4)
    assert torch.allclose(imv2, outs["image"][0], atol=1e-6) and not torch.allclose(titlev2, outs["image"][1], atol=1e-4)
    assert np.abs(outs["text"][1].numpy() - g["feats_text"]).max() < 1e-5


def test_product_fails_loudly_off_gpu_and_in_train_mode():
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    a = A.TINY
    m = HM.PretrainedCLIP(model_type=ClipConfig(**asdict(a))).eval()
    img = A.synth_pixels((1, 3, a.image_resolution, a.image_resolution), 1)
    txt = A.synth_tokens(1, a, 2)
    with pytest.raises(RuntimeError):
        m(img, txt)                       # CPU tensors: no fallback
    m = m.cuda().train()
    with pytest.raises(RuntimeError):
        m(img.cuda(), txt.cuda())


def test_full_size_batch_independence_and_oracle_spot_check():
    """BASELINE sizes (config 2: 256 images + 1536 texts; config 3: 64 eight-frame videos), bf16: the towers run
    on the phased 256x256 GEMM there, which the small cases above never reach.  Size-independent properties:
    (i) an item's embedding does not depend on the batch it is in (first items re-encoded in a small batch, which
    runs on the 128x128 kernel) and (ii) a few items agree with the fp32 oracle within the bf16 tolerance."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd = {}
    sd.update(A.synth_visual(a, 61, prefix="model.visual."))
    sd.update(A.synth_text(a, 62, prefix="model."))
    pv = towers.PackedVision(cuda_sd(sd), "model.visual.", torch.bfloat16)
    pt = towers.PackedText(cuda_sd(sd), "model.", torch.bfloat16, heads=a.transformer_heads)
    img = A.synth_pixels((256, 3, 224, 224), 63)
    txt = A.synth_tokens(1536, a, 64, empty_frac=0.1)
    big_v = pv.forward(img.cuda()).cpu().numpy()
    big_t = pt.forward(txt.cuda()).cpu().numpy()
    assert np.isfinite(big_v).all() and np.isfinite(big_t).all()
    small_v = pv.forward(img[:8].cuda()).cpu().numpy()
    small_t = pt.forward(txt[:24].cuda()).cpu().numpy()
    report("ViT batch independence (256 vs 8)", np.abs(unit(big_v[:8]) - unit(small_v)).max(), 1e-3)
    report("text batch independence (1536 vs 24)", np.abs(unit(big_t[:24]) - unit(small_t)).max(), 1e-3)
    ref_v = CR.encode_image(img[:3], sd, a, "model.visual.").numpy()
    ref_t = CR.encode_text(txt[:6], sd, a, "model.").numpy()
    report("ViT @B=256 vs oracle", np.abs(unit(big_v[:3]) - unit(ref_v)).max(), 1e-3)
    report_text("text @S=1536 vs oracle", unit(big_t[:6]), unit(ref_t), torch.bfloat16, a.embed_dim)
    # config 3: TimeSformer (alt) video tower, 64 videos x 8 frames
    sdv = A.synth_visual(a, 65, nframes=8, prefix="model.visual.")
    pvt = towers.PackedVision(cuda_sd(sdv), "model.visual.", torch.bfloat16)
    vid = A.synth_pixels((64, 8, 3, 224, 224), 66)
    big = pvt.forward(vid.cuda()).cpu().numpy()
    small = pvt.forward(vid[:2].cuda()).cpu().numpy()
    assert np.isfinite(big).all()
    report("TimeSformer batch independence (64 vs 2)", np.abs(unit(big[:2]) - unit(small)).max(), 1e-3)
    ref = T.timesformer_alt(vid[:1], sdv, a, "model.visual.").numpy()
    report("TimeSformer @B=64 vs oracle", np.abs(unit(big[:1]) - unit(ref)).max(), 1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_timesformer_16_frames_vs_oracle(dtype):
    """BASELINE configs[4]: 16-frame TimeSformer (VisualTransformer(nframes=16), timesformer_clip_alt.py:214-250);
    time attention over 16 tokens, 1 + 49*16 tokens per video -- checked against the live oracle."""
    from vtc_amd import towers
    for a, B in ((A.TINY, 3), (A.VIT_B32, 2)):
        sd = A.synth_visual(a, 71, nframes=16, prefix="v.")
        for k in list(sd):                                   # trained temporal_fc is not zero
            if k.endswith("temporal_fc.weight"):
                sd[k] = torch.randn(sd[k].shape, generator=torch.Generator().manual_seed(72)) * 0.02
        x = A.synth_pixels((B, 16, 3, a.image_resolution, a.image_resolution), 73)
        ref = T.timesformer_alt(x, sd, a, "v.").numpy()
        pv = towers.PackedVision(cuda_sd(sd), "v.", dtype)
        out = pv.forward(x.cuda()).cpu().numpy()
        report(f"TimeSformer F=16 {a.vision_width} {dtype}", np.abs(unit(out) - unit(ref)).max(), tol_for(dtype, a.embed_dim))
        report_l2(f"TimeSformer F=16 {a.vision_width} {dtype}", out, ref, dtype)


@pytest.mark.parametrize("dtype", [DTYPES[0], pytest.param(DTYPES[1], marks=pytest.mark.extended)])      # (the 16-bit run repeats test_timesformer_tower_vs_golden through the module class)
def test_tower_module_entry_points_vs_golden(dtype):
    """model.timesformer_clip_alt.VisualTransformer / model.timesformer_clip.VisualTransformer are drop-in modules: the
    reference's constructor, strict state-dict loading and ``forward([B,F,3,H,W])`` (model/timesformer_clip_alt.py:252-286,
    model/timesformer_clip.py:384-438; called directly at timesformer_clip_alt.py:333-360 and as ``self.model.visual(vis)``
    at model/model.py:497,613) reproduce the tower goldens made by the reference's own classes."""
    import model.timesformer_clip as v1
    import model.timesformer_clip_alt as alt
    for prefix, mod, variant in (("tower_alt_", alt, "alt"), ("tower_v1_", v1, "v1")):
        for fname in golden_files(prefix):
            case, g = load_golden(fname)
            a = ARCH[case["arch"]]
            if case["arch"] in ("VIT_B16", "VIT_L14") and not EXTENDED:
                continue        # the same goldens meet the tower in test_timesformer_tower_vs_golden; the module entry point is architecture-blind
            sd = A.synth_visual(a, case["wseed"], nframes=case["nframes"], variant=variant)
            t = mod.VisualTransformer(a.image_resolution, a.vision_patch_size, a.vision_width, a.vision_layers, a.vision_heads,
                                      a.embed_dim, case["nframes"])
            t.load_state_dict(sd, strict=True)
            t = t.eval().cuda()
            t.compute_dtype = dtype
            x = A.synth_pixels((case["B"], case["nframes"], 3, a.image_resolution, a.image_resolution), case["xseed"])
            out = t(x.cuda()).cpu().numpy()
            report(f"{fname} via {mod.__name__}.VisualTransformer {dtype}", np.abs(unit(out) - unit(g["out"])).max(),
                   tol_for(dtype, a.embed_dim))


def test_make_timesformer_factory_runs_forward():
    """make_timesformer_clip_vit_alt(nframes)(x): the factory's product has a working forward (model/__init__.py:21-22)."""
    from vtc_amd.host import clip_arch
    cfg = clip_arch.ClipConfig(**asdict(A.TINY))
    t = clip_arch.make_timesformer_clip_vit_alt(4, clip_model=clip_arch.load(cfg), cfg=cfg).eval().cuda()
    x = A.synth_pixels((2, 4, 3, A.TINY.image_resolution, A.TINY.image_resolution), 3).cuda()
    out = t(x)
    assert out.shape == (2, A.TINY.embed_dim) and torch.isfinite(out).all()
    # the CLIP container's own encode_image / encode_text (upstream API used at model/model.py:332,340)
    m = clip_arch.load(cfg).eval().cuda()
    m.compute_dtype = torch.float32
    m.visual.compute_dtype = torch.float32
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    img = A.synth_pixels((3, 3, A.TINY.image_resolution, A.TINY.image_resolution), 4)
    txt = A.synth_tokens(5, A.TINY, 6, empty_frac=0.2)
    assert (m.encode_image(img.cuda()).cpu() - CR.encode_image(img, sd, A.TINY, "visual.")).abs().max() < 2e-5
    assert (m.encode_text(txt.cuda()).cpu() - CR.encode_text(txt, sd, A.TINY, "")).abs().max() < 2e-5


def test_repack_then_forward_with_overlapping_towers():
    """The weights are re-packed (dtype conversion, transposes, the fp64 temporal fuse) on the caller's stream BEFORE the
    visual tower forks to its side stream: a forward right after a compute_dtype change / an in-place weight update must
    see fully converted weights in BOTH towers (ViT-B/32 size, where the conversions take longer than the first launches)."""
    from vtc_amd.host import model as HM
    a = A.VIT_B32
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type=HM.clip_arch.ClipConfig(**asdict(a)), branch_to_adapt_val="text")
    sd = A.synth_model(a, 5, "timesformer_finaltf", nframes=8)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    B = 4
    vis = A.synth_pixels((B, 8, 3, 224, 224), 8).cuda()
    title = A.synth_tokens(B, a, 9).cuda()
    comments = A.synth_tokens(B * 5, a, 10, empty_frac=0.3).reshape(B, 5, -1).cuda()
    assert m.overlap_towers
    for dtype in (torch.bfloat16, torch.float32, torch.bfloat16):
        m.compute_dtype = dtype                      # forces a repack inside the next forward
        first = [o.clone() for o in m(vis, title, comments)]
        torch.cuda.synchronize()
        again = m(vis, title, comments)              # packed weights now long since converted
        for x, y in zip(first, again):
            assert torch.equal(x, y)
    with torch.no_grad():                            # in-place update (what training does): version bump -> repack
        m.model.text_projection.mul_(1.5)
    first = [o.clone() for o in m(vis, title, comments)]
    torch.cuda.synchronize()
    for x, y in zip(first, m(vis, title, comments)):
        assert torch.equal(x, y)


def test_text_tower_half_layers_statistics():
    """What the IEEE-half text blocks buy: error of the unit-norm text embedding against the fp32 oracle with 0 / 12 half
    layers, 64 sequences of the ViT-B/32 text tower (printed for DESIGN.md; asserted: the shipped setting meets 1e-3)."""
    from vtc_amd import towers
    a = A.VIT_B32
    for wseed in (52, 7):
        sd = A.synth_text(a, wseed, prefix="model.")
        txt = A.synth_tokens(64, a, 54 + wseed, empty_frac=0.25)
        ref = unit(CR.encode_text(txt, sd, a, "model.").numpy())
        for hl in (0, 12):
            pt = towers.PackedText(cuda_sd(sd), "model.", torch.bfloat16, heads=a.transformer_heads, half_layers=hl)
            d = np.abs(unit(pt.forward(txt.cuda()).cpu().numpy()) - ref)
            print(f"[parity] text tower seed {wseed} half_layers={hl}: max {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e}")
            if hl == towers.TEXT_HALF_LAYERS:
                assert d.max() < 1e-3


def test_patch_gather_equals_im2row_path():
    """bf16 pixels take the im2row-free patch GEMM (the LDS-DMA gathers conv1's patches from the pixel tensor,
    timesformer_clip_alt.py:262-274); the same pixels handed over as fp32 go through the im2row matrix.  Same operands,
    same kernel schedule (sizes at which the im2row path runs on the phased kernel too): bit-identical embeddings.
    Covers a tail row tile (15 680 patches = 61.25 tiles), the image tower and the v1 token order; a small batch, where
    the im2row path runs on the 128 x 128 kernel, agrees within bf16 rounding."""
    from vtc_amd import towers
    a = A.VIT_B32
    for nframes, variant, B in ((8, 0, 40), (0, 0, 320), (8, 1, 40), (8, 0, 3)):
        sd = A.synth_visual(a, 91 + nframes, nframes=nframes, prefix="v.", variant="v1" if variant else "alt")
        pv = towers.PackedVision(cuda_sd(sd), "v.", torch.bfloat16)
        shape = (B, nframes, 3, 224, 224) if nframes else (B, 3, 224, 224)
        px = A.synth_pixels(shape, 94).bfloat16()
        got = pv.forward(px.cuda())
        ref = pv.forward(px.float().cuda())
        if B >= 40:
            assert torch.equal(got, ref), (nframes, variant, (got - ref).abs().max().item())
        else:
            report("patch gather vs im2row, small batch", np.abs(unit(got.cpu().numpy()) - unit(ref.cpu().numpy())).max(), 1e-3)


@pytest.mark.extended      # (default run: test_last_block_on_the_output_rows_only_... holds both switch positions of the same four towers to the oracle)
def test_folded_layernorm_and_layernorm_kernels_agree_with_the_oracle():
    """16-bit modes fold each LayerNorm into the projection behind it (include/vtc_hip.h vtc_block_w *_wf/_s/_c;
    LN(x) W^T + b = rstd (x (g.W)^T - mean s) + c; model/timesformer_clip_alt.py:142-175 ln_time / ln_1 / ln_2).  Both
    switch positions stay within the bf16 tolerance of the fp32 oracle, for the alt and v1 video towers, the image tower
    and the (ragged and dense) text tower; odd batch sizes exercise the padded rows."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd_alt = A.synth_visual(a, 101, nframes=8, prefix="v.")
    for k in list(sd_alt):
        if k.endswith("temporal_fc.weight"):
            sd_alt[k] = torch.randn(sd_alt[k].shape, generator=torch.Generator().manual_seed(102)) * 0.02
    sd_v1 = A.synth_visual(a, 103, nframes=8, prefix="v.", variant="v1")
    sd_img = A.synth_visual(a, 104, prefix="v.")
    sd_txt = A.synth_text(a, 105, prefix="t.")
    vid = A.synth_pixels((3, 8, 3, 224, 224), 106)
    img = A.synth_pixels((5, 3, 224, 224), 107)
    txt = A.synth_tokens(7, a, 108, empty_frac=0.2)
    refs = {
        "alt": T.timesformer_alt(vid, sd_alt, a, "v.").numpy(),
        "v1": T.timesformer_v1(vid, sd_v1, a, "v.").numpy(),
        "img": CR.encode_image(img, sd_img, a, "v.").numpy(),
        "txt": CR.encode_text(txt, sd_txt, a, "t.").numpy(),
    }
    packed = {
        "alt": (towers.PackedVision(cuda_sd(sd_alt), "v.", torch.bfloat16), vid),
        "v1": (towers.PackedVision(cuda_sd(sd_v1), "v.", torch.bfloat16), vid),
        "img": (towers.PackedVision(cuda_sd(sd_img), "v.", torch.bfloat16), img),
        "txt": (towers.PackedText(cuda_sd(sd_txt), "t.", torch.bfloat16, heads=a.transformer_heads), txt),
    }
    outs = {}
    for on in (1, 0):
        for name, (pk, x) in packed.items():
            pk.w.flags = towers.tower_flags(ln_fold=bool(on))
            outs[(name, on)] = pk.forward(x.cuda()).cpu().numpy()
            if name == "txt":
                outs[("txt_dense", on)] = pk.forward(x.cuda(), ragged=False).cpu().numpy()
    for (name, on), got in outs.items():
        ref = refs.get(name.split("_")[0])
        if ref is not None:
            report(f"{name} fold={on} vs oracle", np.abs(unit(got) - unit(ref)).max(), 1e-3)
    for name in ("alt", "v1", "img", "txt", "txt_dense"):
        d = np.abs(unit(outs[(name, 1)]) - unit(outs[(name, 0)])).max()
        assert 0 < d < 1e-3, (name, d)          # two rounding regimes of the same tower: different bits, same embedding


@pytest.mark.parametrize("dtype", [pytest.param(DTYPES[0], marks=pytest.mark.extended), DTYPES[1]])      # (fp32 has no fold / LayerNorm-kernel alternative: one configuration, extended)
def test_last_block_on_the_output_rows_only_equals_the_full_last_block(dtype):
    """Behind the last block a tower reads one row per item (x[:, 0] -> ln_post, model/timesformer_clip_alt.py:281,
    model/timesformer_clip.py:433; the EOT row -> ln_final), so that block's out_proj + MLP run on those rows only by default
    (towers.hip last_block_tail).  VTC_TOWER_FULL_LAST_LAYER computes every row as the reference does: same embeddings -- fp32
    to summation order, bf16 within the tolerance both hold against the oracle -- for the alt / v1 video towers, the image tower
    and the ragged, dense and two-array text tower; folded and LayerNorm-kernel paths; odd batch sizes."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd_alt = A.synth_visual(a, 201, nframes=8, prefix="v.")
    sd_v1 = A.synth_visual(a, 203, nframes=8, prefix="v.", variant="v1")
    sd_img = A.synth_visual(a, 204, prefix="v.")
    sd_txt = A.synth_text(a, 205, prefix="t.")
    vid = A.synth_pixels((3, 8, 3, 224, 224), 206)
    img = A.synth_pixels((5, 3, 224, 224), 207)
    txt = A.synth_tokens(7, a, 208, empty_frac=0.2)
    refs = {
        "alt": T.timesformer_alt(vid, sd_alt, a, "v.").numpy(),
        "v1": T.timesformer_v1(vid, sd_v1, a, "v.").numpy(),
        "img": CR.encode_image(img, sd_img, a, "v.").numpy(),
        "txt": CR.encode_text(txt, sd_txt, a, "t.").numpy(),
    }
    packed = {
        "alt": (towers.PackedVision(cuda_sd(sd_alt), "v.", dtype), vid),
        "v1": (towers.PackedVision(cuda_sd(sd_v1), "v.", dtype), vid),
        "img": (towers.PackedVision(cuda_sd(sd_img), "v.", dtype), img),
        "txt": (towers.PackedText(cuda_sd(sd_txt), "t.", dtype, heads=a.transformer_heads), txt),
    }
    tol = tol_for(dtype)
    from vtc_amd import _lib as L
    lib = L.lib()
    # fold 1 / 0: folded LayerNorm / LayerNorm kernels, both with the last block's queries pruned as well (K and V projected for every
    # row, one query per sequence: sq_attn_kernel)
    for fold in ((1, 0) if dtype == torch.bfloat16 else (1,)):
        outs, launches = {}, {}
        for full in (0, 1):
            for name, (pk, x) in packed.items():
                pk.w.flags = towers.tower_flags(ln_fold=bool(fold), full_last_layer=bool(full))
                n0 = lib.vtc_debug_launch_count()
                outs[(name, full)] = pk.forward(x.cuda()).cpu().numpy()
                launches[(name, full)] = lib.vtc_debug_launch_count() - n0
                if name == "txt":
                    outs[("txt_dense", full)] = pk.forward(x.cuda(), ragged=False).cpu().numpy()
                    both = pk.forward(x[:4].cuda(), ids_b=x[4:].cuda()).cpu().numpy()
                    outs[("txt_two", full)] = both
        for (name, full), got in outs.items():
            report(f"{name} fold={fold} full_last={full} vs oracle", np.abs(unit(got) - unit(refs[name.split('_')[0]])).max(), tol)
        for name in ("alt", "v1", "img", "txt", "txt_dense", "txt_two"):
            d = np.abs(unit(outs[(name, 0)]) - unit(outs[(name, 1)])).max()
            print(f"[parity] {name} fold={fold}: output rows only vs full last block {d:.3e}")
            assert d < (2e-6 if dtype == torch.float32 else tol), (name, fold, d)
        for pk, _ in packed.values():
            pk.w.flags = towers.DEFAULT_FLAGS


@pytest.mark.parametrize("dtype", DTYPES)
def test_single_block_towers_and_single_items_through_the_last_block_tail(dtype):
    """The last block is also the FIRST one (layers = 1: the residual stream reaches the tail without ever having been through a
    residual GEMM), width 256 (the smallest that folds the LayerNorms), one item per call: alt / v1 video towers, image tower
    and text tower against the oracle."""
    from dataclasses import replace
    from vtc_amd import towers
    a = replace(A.TINY, embed_dim=128, vision_layers=1, vision_width=256, transformer_width=256, transformer_heads=4, transformer_layers=1)
    sd_alt = A.synth_visual(a, 301, nframes=4, prefix="v.")
    for k in list(sd_alt):
        if k.endswith("temporal_fc.weight"):
            sd_alt[k] = torch.randn(sd_alt[k].shape, generator=torch.Generator().manual_seed(302)) * 0.02
    sd_v1 = A.synth_visual(a, 303, nframes=4, prefix="v.", variant="v1")
    sd_img = A.synth_visual(a, 304, prefix="v.")
    sd_txt = A.synth_text(a, 305, prefix="t.")
    tol = tol_for(dtype, a.embed_dim)
    for n in (1, 3):
        vid = A.synth_pixels((n, 4, 3, 64, 64), 306 + n)
        img = A.synth_pixels((n, 3, 64, 64), 307 + n)
        txt = A.synth_tokens(n, a, 308 + n)
        cases = [("alt", towers.PackedVision(cuda_sd(sd_alt), "v.", dtype), vid, T.timesformer_alt(vid, sd_alt, a, "v.")),
                 ("v1", towers.PackedVision(cuda_sd(sd_v1), "v.", dtype), vid, T.timesformer_v1(vid, sd_v1, a, "v.")),
                 ("img", towers.PackedVision(cuda_sd(sd_img), "v.", dtype), img, CR.encode_image(img, sd_img, a, "v.")),
                 ("txt", towers.PackedText(cuda_sd(sd_txt), "t.", dtype, heads=a.transformer_heads), txt, CR.encode_text(txt, sd_txt, a, "t."))]
        for name, pk, x, ref in cases:
            for full in (0, 1):
                pk.w.flags = towers.tower_flags(full_last_layer=bool(full))
                got = pk.forward(x.cuda()).cpu().numpy()
                report(f"1-block {name} n={n} full_last={full}", np.abs(unit(got) - unit(ref.numpy())).max(), tol)


def test_folded_layernorm_is_insensitive_to_a_row_mean():
    """The folded LayerNorm rounds x itself (not LN(x)) to the operand format, so a row mean large against the row's spread
    would cost precision -- the (hi, lo) stream is therefore stored centred (every reader is a LayerNorm: a per-row constant
    is invisible to it; gemm.hip SPLIT, norm.hip cast_rowstats_kernel).  ln_pre.bias + 10 puts |mean| / std at ~10 on every
    row of every layer's input: the embedding must stay within the bf16 tolerance of the fp32 oracle."""
    from vtc_amd import towers
    a = A.VIT_B32
    img = A.synth_pixels((4, 3, 224, 224), 7)
    sd = A.synth_visual(a, 104, prefix="v.")
    sd["v.ln_pre.bias"] = sd["v.ln_pre.bias"] + 10.0
    ref = CR.encode_image(img, sd, a, "v.").numpy()
    got = towers.PackedVision(cuda_sd(sd), "v.", torch.bfloat16).forward(img.cuda()).cpu().numpy()
    report("image tower, DC offset 10 on the residual rows", np.abs(unit(got) - unit(ref)).max(), 1e-3)


def test_folded_layernorm_across_an_operand_format_boundary():
    """Text tower with only the first 6 blocks on IEEE-half operands (vtc_text_w.half_layers = 6): at the boundary the (hi, lo)
    stream changes format -- merged back to fp32 and re-cast (towers.hip ln_proj).  Ragged and dense, against the fp32 oracle."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_text(a, 111, prefix="t.")
    txt = A.synth_tokens(9, a, 112, empty_frac=0.2)
    ref = CR.encode_text(txt, sd, a, "t.").numpy()
    for hl in (6, 0, 12):
        pk = towers.PackedText(cuda_sd(sd), "t.", torch.bfloat16, heads=a.transformer_heads, half_layers=hl)
        for ragged in (True, False):
            got = pk.forward(txt.cuda(), ragged=ragged).cpu().numpy()
            # all-bf16 blocks sit at the bf16 floor of this tower: 2e-3 there, 1e-3 with any half blocks in front
            report(f"text half_layers={hl} ragged={ragged}", np.abs(unit(got) - unit(ref)).max(), 2e-3 if hl == 0 else 1.5e-3 if hl == 6 else 1e-3)


def test_folded_layernorm_odd_batches_agree_with_the_layernorm_kernels():
    """Odd batch sizes (1 row block and many, rows far from a multiple of 256: the padded tiles) through the folded and the
    LayerNorm-kernel paths of the same packed towers: the two roundings of one embedding differ by a fraction of the tolerance,
    and an item's embedding does not depend on the batch around it."""
    from vtc_amd import towers
    a = A.VIT_B32
    pv = towers.PackedVision(cuda_sd(A.synth_visual(a, 121, nframes=8, prefix="v.")), "v.", torch.bfloat16)
    pi = towers.PackedVision(cuda_sd(A.synth_visual(a, 122, prefix="v.")), "v.", torch.bfloat16)
    pt = towers.PackedText(cuda_sd(A.synth_text(a, 123, prefix="t.")), "t.", torch.bfloat16, heads=a.transformer_heads)
    vid = A.synth_pixels((5, 8, 3, 224, 224), 124).cuda()
    img = A.synth_pixels((33, 3, 224, 224), 125).cuda()
    txt = A.synth_tokens(50, a, 126, empty_frac=0.2).cuda()
    cases = [(pv, vid, (1, 2, 5)), (pi, img, (1, 7, 33)), (pt, txt, (1, 13, 50))]
    outs = {}
    for on in (1, 0):
        for ci, (pk, x, sizes) in enumerate(cases):
            pk.w.flags = towers.tower_flags(ln_fold=bool(on))
            for n in sizes:
                outs[(ci, n, on)] = unit(pk.forward(x[:n]).cpu().numpy())
    for ci, (pk, x, sizes) in enumerate(cases):
        for n in sizes:
            d = np.abs(outs[(ci, n, 1)] - outs[(ci, n, 0)]).max()
            assert np.isfinite(outs[(ci, n, 1)]).all() and d < 8e-4, (ci, n, d)
            # batch independence of the folded path: the first item alone vs inside the batch (different tile walks, same rounding)
            d1 = np.abs(outs[(ci, n, 1)][:1] - outs[(ci, sizes[0], 1)][:1]).max()
            assert d1 < 8e-4, (ci, n, d1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_device_side_ragged_bookkeeping_and_two_id_arrays(dtype):
    """vtc_text_forward2: EOT positions (first maximum id, as upstream's `text.argmax(-1)`), their prefix sums and the row count
    are computed on the device and every kernel reads the count there.  Must be BIT-identical to the host-bookkeeping path of
    round 2 (vtc_text_forward_ragged: same rows, same tiles -- the grids are merely sized for the dense upper bound), for one
    id array and for titles + comments handed over as two arrays (== their concatenation); the dense path over two arrays
    equals the dense path over the concatenation.  Sizes: a single sequence, a batch below one 256-row tile, and one with
    several hundred sequences (more than one scan chunk would need > 1024: covered by 1 300)."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_text(a, 152, prefix="model.")
    pt = towers.PackedText(cuda_sd(sd), "model.", dtype, heads=a.transformer_heads)
    for S in (1, 7, 1300):
        txt = A.synth_tokens(S, a, 154 + S, empty_frac=0.25)
        if S > 5:
            txt[3, 1:76] = torch.randint(1, A.SOT, (75,)); txt[3, 76] = A.EOT          # full length
            txt[5] = torch.randint(1, 1000, (77,)); txt[5, 40] = 48000; txt[5, 60] = 48000   # no EOT, a repeated maximum: the first wins
        t = txt.cuda()
        host = pt.forward_host_offsets(t)
        dev = pt.forward(t, ragged=True)
        assert torch.isfinite(dev).all() and torch.equal(dev, host), S
        if S > 1:
            k = S // 3 + 1
            two = pt.forward(t[:k].contiguous(), ragged=True, ids_b=t[k:].contiguous())
            assert torch.equal(two, host), S
            dense = pt.forward(t, ragged=False)
            dense2 = pt.forward(t[:k].contiguous(), ragged=False, ids_b=t[k:].contiguous())
            assert torch.equal(dense, dense2), S


def test_config3_forward_is_free_of_host_syncs_and_torch_kernels():
    """no torch compute and no host sync inside the forward.  A config-3 forward (TimeSformer video + titles +
    comments + CAM + similarity, ragged text) runs under torch.cuda.set_sync_debug_mode("error"): any .item() / blocking copy /
    synchronisation raises.  The first call packs the weights (host-side conversion + uploads, which do synchronise) and sizes
    the workspaces; the checked call is the steady state."""
    from vtc_amd.host import model as HM
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval().cuda()
    m.compute_dtype = torch.bfloat16
    a = A.VIT_B32
    vid = A.synth_pixels((3, 8, 3, 224, 224), 201).cuda().bfloat16()
    title = A.synth_tokens(3, a, 202).cuda()
    comments = A.synth_tokens(15, a, 203, empty_frac=0.3).reshape(3, 5, -1).cuda()
    ref = m(vid, title, comments)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = m(vid, title, comments)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    for x, y in zip(out, ref):
        assert torch.equal(x, y)


def test_half_text_blocks_under_range_stress_and_overflow_fallback():
    """the bf16 mode runs the text blocks on IEEE-half operands; half has 5 exponent bits.
    (a) stress inside the range: an outlier channel planted in token_embedding (residual stream ~30 = 100x its usual size on that
        channel, the 'massive activation' pattern of real checkpoints) and ln_2.weight x 30 on the first four blocks (c_fc
        pre-activations and MLP hidden values in the hundreds to thousands) -- the embedding must stay within 1e-3 of the fp32
        oracle, ragged and dense;
    (b) a token whose embedding leaves the half range (1e5 > 65504): the FIRST forward after packing detects the non-finite
        output synchronously, re-packs the blocks as bf16 and recomputes (finite, warning, `range_fallbacks`);
    (c) the same token arriving in a LATER batch: that call returns NaN rows (nothing synchronises), the next call sees the
        flag in pinned host memory, switches with a warning, and is finite."""
    import warnings
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_text(a, 211, prefix="t.")
    sd["t.token_embedding.weight"][:, 7] += 30.0
    for l in range(4):
        sd[f"t.transformer.resblocks.{l}.ln_2.weight"] *= 30.0
    txt = A.synth_tokens(24, a, 212, empty_frac=0.2)
    ref = unit(CR.encode_text(txt, sd, a, "t.").numpy())
    pt = towers.PackedText(cuda_sd(sd), "t.", torch.bfloat16, heads=a.transformer_heads)
    for ragged in (True, False):
        got = pt.forward(txt.cuda(), ragged=ragged)
        assert torch.isfinite(got).all() and pt.range_fallbacks == 0 and pt.w.half_layers == 12
        report(f"half text blocks under range stress, ragged={ragged}", np.abs(unit(got.cpu().numpy()) - ref).max(), 1e-3)
    # (b) overflow on the first forward
    sd2 = A.synth_text(a, 213, prefix="t.")
    sd2["t.token_embedding.weight"][777, 3] = 1.0e5
    txt2 = A.synth_tokens(12, a, 214)
    txt2[5, 2] = 777
    ref2 = unit(CR.encode_text(txt2, sd2, a, "t.").numpy())
    pt2 = towers.PackedText(cuda_sd(sd2), "t.", torch.bfloat16, heads=a.transformer_heads)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        got2 = pt2.forward(txt2.cuda())
    assert pt2.range_fallbacks == 1 and pt2.w.half_layers == 0 and any("IEEE-half range" in str(w.message) for w in wlist)
    assert torch.isfinite(got2).all()
    d2 = np.abs(unit(got2.cpu().numpy()) - ref2)
    print(f"[parity] bf16 fallback after a half overflow: max {d2.max():.3e} (rows without the outlier token: {np.delete(d2, 5, 0).max():.3e})")
    assert np.delete(d2, 5, 0).max() < 2e-3          # all-bf16 blocks: the floor of DESIGN.md 2
    # (c) overflow in a later batch
    pt3 = towers.PackedText(cuda_sd(sd2), "t.", torch.bfloat16, heads=a.transformer_heads)
    clean = A.synth_tokens(12, a, 215)
    clean[clean == 777] = 778
    assert torch.isfinite(pt3.forward(clean.cuda())).all() and pt3.range_fallbacks == 0
    bad = pt3.forward(txt2.cuda())
    torch.cuda.synchronize()
    assert not torch.isfinite(bad[5]).all() and torch.isfinite(bad[:5]).all()      # only the sequence with the outlier token
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        again = pt3.forward(txt2.cuda())
    assert pt3.range_fallbacks == 1 and any("earlier forward" in str(w.message) for w in wlist) and torch.isfinite(again).all()


def test_split_k_c_proj_at_batch_1_and_2_equals_the_one_pass_gemm():
    """at batch 1 - 2 (<= 1024 padded rows) the MLP's c_proj (K = 3072 / 2048 on a handful of 64 x 64 tiles) runs split
    over K, and ONE row pass sums the slices, applies the residual update on the (hi, lo) stream and writes the LayerNorm statistics
    (no fold_stats launch).  Same embeddings as the one-pass GEMM (VTC_TOWER_NO_SPLITK) up to the summation order over K -- which, through
    the bf16 roundings downstream, means: to the 16-bit tower's noise level, both equally close to the oracle (1e-3) --, the same launch
    count (GEMM + fold_stats -> split GEMM + row pass); a batch too large for it
    (5 videos: 2 048 padded rows) takes the one-pass path whatever the flag."""
    from vtc_amd import _lib as L
    from vtc_amd import towers
    a = A.VIT_B32
    lib = L.lib()
    sdv = A.synth_visual(a, 301, nframes=8, prefix="v.")
    for k in list(sdv):
        if k.endswith("temporal_fc.weight"):
            sdv[k] = torch.randn(sdv[k].shape, generator=torch.Generator().manual_seed(302)) * 0.02
    sdt = A.synth_text(a, 303, prefix="t.")
    vid = A.synth_pixels((5, 8, 3, 224, 224), 304)
    txt = A.synth_tokens(12, a, 305, empty_frac=0.2)
    ref_v = T.timesformer_alt(vid[:2], sdv, a, "v.").numpy()
    ref_t = CR.encode_text(txt, sdt, a, "t.").numpy()

    def run(tower, x, flags):
        tower.w.flags = flags
        tower.forward(x)
        n0 = lib.vtc_debug_launch_count()
        out = tower.forward(x)
        return out.cpu().numpy(), lib.vtc_debug_launch_count() - n0

    pv = towers.PackedVision(cuda_sd(sdv), "v.", torch.bfloat16)
    pt = towers.PackedText(cuda_sd(sdt), "t.", torch.bfloat16, heads=a.transformer_heads)
    base = towers.tower_flags()
    assert base & L.TOWER_NO_SPLITK == 0
    for B in (1, 2):
        split, n_split = run(pv, vid[:B].cuda(), base)
        one, n_one = run(pv, vid[:B].cuda(), base | L.TOWER_NO_SPLITK)
        # two roundings of ONE computation: an fp32-level difference in a residual row (summation order over K) flips bf16 roundings of the
        # operand copy further down, so the two agree to the 16-bit tower's noise level, not to 1e-7 -- the bound of the batch-independence
        # tests; what must hold is that BOTH sit equally close to the oracle
        report(f"video tower B={B}: split-K c_proj vs one-pass", np.abs(unit(split) - unit(one)).max(), 1e-3)
        e_split, e_one = np.abs(unit(split) - unit(ref_v[:B])).max(), np.abs(unit(one) - unit(ref_v[:B])).max()
        report(f"video tower B={B}: split-K c_proj vs oracle (one-pass: {e_one:.3e})", e_split, 1e-3)
        assert e_split < 2.0 * e_one + 1e-4
        assert n_one - n_split == 0 and n_split > 100, (n_split, n_one)       # 11 x (GEMM + fold_stats) -> 11 x (split GEMM + row pass)
    big_split, _ = run(pv, vid.cuda(), base)
    big_one, _ = run(pv, vid.cuda(), base | L.TOWER_NO_SPLITK)
    assert np.array_equal(big_split, big_one)                                  # 5 videos = 2 048 padded rows: no split either way
    for ragged in (True, False):
        pt.w.flags = base
        s_ = pt.forward(txt.cuda(), ragged=ragged).cpu().numpy()
        pt.w.flags = base | L.TOWER_NO_SPLITK
        o_ = pt.forward(txt.cuda(), ragged=ragged).cpu().numpy()
        report(f"text tower 12 sequences ragged={ragged}: split-K c_proj vs one-pass", np.abs(unit(s_) - unit(o_)).max(), 1e-3)
        report_text(f"text tower 12 sequences ragged={ragged}: split-K vs oracle", unit(s_), unit(ref_t), torch.bfloat16, a.embed_dim)


def _stress_visual(a, seed, nframes, variant, s_attn, s_mlp):
    """A video tower with the activation statistics of trained checkpoints instead of the tame init-style ones:
      * a x100 outlier channel in class_embedding and in positional_embedding, and a 'massive activation' channel that SURVIVES ln_pre
        (ln_pre.bias[7] = 30: the residual stream carries ~30 on that channel in every row, 100x its usual size -- the pattern of real
        ViT checkpoints; the text-tower stress plants the same in token_embedding);
      * ln_2.weight x s_mlp on the first four blocks (c_fc pre-activations in the tens to hundreds, MLP hidden values in the hundreds to
        thousands) and ln_1 / ln_time .weight x s_attn there (q and k BOTH scale: attention logits x s_attn^2, sharper softmaxes);
      * a heavy-tailed patch embedding: conv1.weight scaled per output channel by a log-normal (sigma 1) factor;
      * trained-like (non-zero) temporal_fc.
    How far this can go is a property of the NETWORK, not of an implementation: at s_attn = 10 the softmaxes are near one-hot and the
    reference's own fp32 arithmetic disagrees with fp64 by 1.4e-4 (alt) / 1.3e-3 (v1) on the unit-norm embedding; at 30 by 7e-3 / 3e-2
    (measured with the oracle).  The test therefore measures that conditioning and only asks 1e-3 where the problem is well posed."""
    sd = A.synth_visual(a, seed, nframes=nframes, prefix="v.", variant=variant)
    g = torch.Generator().manual_seed(seed + 1)
    W = a.vision_width
    sd["v.class_embedding"][11] *= 100.0
    sd["v.positional_embedding"][:, 13] *= 100.0
    sd["v.ln_pre.bias"][7] += 30.0
    for l in range(4):
        for ln, sc in (("ln_1", s_attn), ("ln_time", s_attn), ("ln_2", s_mlp)):
            k = f"v.transformer.resblocks.{l}.{ln}.weight"
            if k in sd:
                sd[k] *= sc
    sd["v.conv1.weight"] *= torch.exp(torch.randn(W, generator=g))[:, None, None, None]
    for k in list(sd):
        if k.endswith("temporal_fc.weight"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    return sd


FP32_VS_FP64_TAME = 8e-8      # the oracle's fp32 vs fp64 unit-norm embedding on init-style weights (alt and v1, F = 8): the baseline conditioning


@pytest.mark.parametrize("variant,F,s_attn", [("alt", 8, 3.0), ("alt", 16, 3.0), ("v1", 8, 3.0), ("v1", 16, 3.0), ("alt", 8, 6.0), ("v1", 8, 6.0)])
def test_video_tower_16bit_under_massive_activations(variant, F, s_attn):
    """the 16-bit video tower -- bf16 operands, (hi, lo) row-centred residual stream, folded LayerNorm -- was held to 1e-3
    on init-style weights only.  Here: ViT-B/32 TimeSformer (model/timesformer_clip_alt.py and the v1 variant model/timesformer_clip.py),
    F = 8 and 16 frames, under the stress of _stress_visual, against the oracle in fp64.  What the measurements say (round 6):
      * the conditioning of a case is MEASURED: kappa = (oracle fp32 vs oracle fp64) / (the same on init-style weights) -- 1.6 - 2.1 at
        s_attn = 3, 4 (alt) / 27 (v1) at s_attn = 6;
      * the bf16 mode's error scales with it: 3.8e-4 on init-style weights (kappa 1), 7.3e-4 - 1.07e-3 at kappa ~ 2, 1.7e-3 at 4, 9e-3 at
        27.  So BASELINE's 1e-3 is a statement about init-style conditioning; on a network whose own fp32 arithmetic is kappa times less
        stable, bf16 operands (8 significant bits) are kappa times noisier -- asserted: kappa x 1e-3;
      * the IEEE-half mode (compute_dtype = torch.float16, round 6: 11 significant bits at the same MFMA rate and bytes, range +-65504)
        stays within a FLAT 1e-3 at kappa <= 4 -- the mode to choose for checkpoints with heavy activation statistics;
      * fp32 mode: kappa x 1e-5 (which also pins that the stress itself is computed right)."""
    from vtc_amd import towers
    a = A.VIT_B32
    sd = _stress_visual(a, 231 + F, F, variant, s_attn, 30.0)
    x = A.synth_pixels((2, F, 3, 224, 224), 233).bfloat16().float()          # bf16-representable pixels: both sides see the same input
    oracle = T.timesformer_alt if variant == "alt" else T.timesformer_v1
    ref32 = oracle(x, sd, a, "v.").double().numpy()
    ref = oracle(x.double(), {k: v.double() for k, v in sd.items()}, a, "v.").numpy()
    assert np.isfinite(ref).all()
    kappa = max(1.0, float(np.abs(unit(ref32) - unit(ref)).max()) / FP32_VS_FP64_TAME)
    print(f"[parity] video tower {variant} F={F} s_attn={s_attn}: oracle fp32 vs fp64 {np.abs(unit(ref32) - unit(ref)).max():.2e} -> kappa {kappa:.1f}; "
          f"cosine between the two items {float((unit(ref)[0] * unit(ref)[1]).sum()):.4f}")
    if s_attn <= 3.0:
        assert kappa < 4.0, "the 'well posed' stress level is not: re-calibrate"
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        pv = towers.PackedVision(cuda_sd(sd), "v.", dtype)
        assert pv.w.variant == (0 if variant == "alt" else 1)
        out = pv.forward(x.cuda() if dtype == torch.float32 else x.cuda().to(dtype)).cpu().numpy().astype(np.float64)
        assert np.isfinite(out).all()
        tol = kappa * 1e-5 if dtype == torch.float32 else kappa * 1e-3 if dtype == torch.bfloat16 else (1e-3 if kappa <= 5.0 else kappa * 2.5e-4)
        report(f"video tower {variant} F={F} s_attn={s_attn} kappa={kappa:.1f} {dtype} under massive activations", np.abs(unit(out) - unit(ref)).max(), tol)


def test_video_and_image_towers_in_ieee_half_mode_vs_oracle():
    """Round 6: the vision towers take IEEE-half operands too (compute_dtype = torch.float16 / VTC_COMPUTE_DTYPE=f16 / --dtype f16): the
    same kernels on f16_t -- the bf16 mode already runs the text blocks on them -- with a half (hi, lo) residual stream.  On init-style
    weights: ViT-B/32 image tower, TimeSformer alt and v1 at F = 8 against the fp32 oracle; bf16 pixels, half pixels, uint8 pixels."""
    from vtc_amd import towers
    a = A.VIT_B32
    for name, sd, x, oracle in (
            ("ViT-B/32 image", A.synth_visual(a, 51, prefix="v."), A.synth_pixels((3, 3, 224, 224), 53), lambda x_, sd_: CR.encode_image(x_, sd_, a, "v.")),
            ("TimeSformer alt F=8", A.synth_visual(a, 65, nframes=8, prefix="v."), A.synth_pixels((2, 8, 3, 224, 224), 66), lambda x_, sd_: T.timesformer_alt(x_, sd_, a, "v.")),
            ("TimeSformer v1 F=8", A.synth_visual(a, 67, nframes=8, prefix="v.", variant="v1"), A.synth_pixels((2, 8, 3, 224, 224), 68),
             lambda x_, sd_: T.timesformer_v1(x_, sd_, a, "v."))):
        x = x.half().float()                                  # half-representable pixels
        ref = oracle(x, sd).numpy()
        errs = {}
        for dtype in (torch.bfloat16, torch.float16):
            pv = towers.PackedVision(cuda_sd(sd), "v.", dtype)
            errs[dtype] = np.abs(unit(pv.forward(x.cuda().to(dtype)).cpu().numpy()) - unit(ref)).max()
        pvh = towers.PackedVision(cuda_sd(sd), "v.", torch.float16)
        e32 = np.abs(unit(pvh.forward(x.cuda()).cpu().numpy()) - unit(ref)).max()           # fp32 pixels in: the im2row path casts
        print(f"[parity] {name}: bf16 mode {errs[torch.bfloat16]:.3e}, IEEE-half mode {errs[torch.float16]:.3e} (fp32 pixels in: {e32:.3e}) vs the fp32 oracle")
        assert errs[torch.float16] < 2.5e-4 and e32 < 2.5e-4 and errs[torch.bfloat16] < 1e-3
        assert errs[torch.float16] < errs[torch.bfloat16]


def test_half_mode_vision_tower_falls_back_to_bf16_when_the_range_is_left():
    """The IEEE-half mode's range guard (as the text tower's): a residual-stream value beyond +-65504 (ln_pre.bias[7] = 1e5) makes the
    output non-finite; the FIRST forward after packing notices synchronously, re-packs the tower as bf16 operands, warns and recomputes
    (finite); an overflow that first occurs in a LATER batch returns NaN rows for that call and switches at the next."""
    import warnings
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_visual(a, 341, prefix="v.")
    sd["v.ln_pre.bias"][7] = 1.0e5
    x = A.synth_pixels((2, 3, 224, 224), 342)
    pv = towers.PackedVision(cuda_sd(sd), "v.", torch.float16)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        out = pv.forward(x.cuda())
    assert pv.range_fallbacks == 1 and pv.dtype == torch.bfloat16 and any("IEEE-half range" in str(w.message) for w in wl)
    assert torch.isfinite(out).all()
    ref = CR.encode_image(x, sd, a, "v.").numpy()
    print(f"[parity] half-mode vision tower after the bf16 fallback (a 1e5 channel in the stream): max err {np.abs(unit(out.cpu().numpy()) - unit(ref)).max():.3e}")
    # a later batch: pixels that only then push a value out of range (a tame tower, one pixel plane scaled by 3e4)
    sd2 = A.synth_visual(a, 343, prefix="v.")
    pv2 = towers.PackedVision(cuda_sd(sd2), "v.", torch.float16)
    assert torch.isfinite(pv2.forward(x.cuda())).all() and pv2.range_fallbacks == 0
    big = x.clone()
    big[1] *= 3.0e4
    bad = pv2.forward(big.cuda())
    torch.cuda.synchronize()
    if torch.isfinite(bad).all():
        pytest.skip("ln_pre tamed the scaled pixels: no overflow to test on this tower")
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        again = pv2.forward(big.cuda())
    assert pv2.range_fallbacks == 1 and pv2.dtype == torch.bfloat16 and torch.isfinite(again).all()
    assert any("earlier forward" in str(w.message) for w in wl)


def test_config3_wrapper_in_ieee_half_mode_vs_oracle(monkeypatch):
    """compute_dtype = torch.float16 through the drop-in wrapper (config 3: TimeSformer + title + comments + CAM), ViT-B/32, B = 3: both
    embedding sets and the cosine similarity against the fp32 oracle, beside the bf16 mode on the same inputs; VTC_COMPUTE_DTYPE=f16 and
    the name table select it."""
    from vtc_amd.host import model as HM
    a = A.VIT_B32
    sd = A.synth_model(a, 321, "timesformer_finaltf", nframes=8)
    g = torch.Generator().manual_seed(322)
    for k in list(sd):
        if k.endswith("temporal_fc.weight") or (k.startswith("final_transformer.") and (k.endswith("out_proj.weight") or k.endswith("c_proj.weight"))):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    monkeypatch.setenv("VTC_COMPUTE_DTYPE", "f16")
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
    assert m.compute_dtype == torch.float16 and HM.parse_compute_dtype("half") == torch.float16
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    vis = A.synth_pixels((3, 8, 3, 224, 224), 323).half().float()
    title = A.synth_tokens(3, a, 324)
    comments = A.synth_tokens(15, a, 325, empty_frac=0.3).reshape(3, 5, -1)
    ref = M.pretrained_clip_timesformer_finaltf(vis, title, comments, sd, a, "text")
    errs = {}
    for dt in (torch.float16, torch.bfloat16):
        m.compute_dtype = dt
        out = m(vis.cuda().to(dt), title.cuda(), comments.cuda())
        m.check_finite()
        errs[dt] = [float((o.cpu() - r).abs().max()) for o, r in zip(out[:2], ref[:2])]
    print(f"[parity] config-3 wrapper, B=3: IEEE-half mode feats_vis / feats_text {errs[torch.float16][0]:.3e} / {errs[torch.float16][1]:.3e}; "
          f"bf16 mode {errs[torch.bfloat16][0]:.3e} / {errs[torch.bfloat16][1]:.3e} (tol 1e-3)")
    assert max(errs[torch.float16]) < 3e-4 and max(errs[torch.bfloat16]) < 1e-3 and errs[torch.float16][0] < errs[torch.bfloat16][0]


def test_one_launch_cam_equals_the_multi_launch_path():
    """cam.hip: small batches run the whole Context Adapter Module (model/model.py:141-214: token build with the empty-comment
    mask, two transformer layers over the 1 + nc tokens of an item, avg-of-normalised, residual activation, final normalise) as
    ONE cooperative launch, spread over the CUs by output columns with grid barriers between the phases.  Same fp32
    arithmetic up to summation order as the ~30 generic launches (vtc_cam_w.flags = VTC_CAM_NO_FUSED): <= 2e-6 on unit
    vectors, for every residual activation, trained-like (non-zero) projections, 1 / 3 / 50 / 85 items, and 2 comments instead
    of 5; the goldens of the wrappers (wrap_03..09,14,15: B = 2..4) pin both against the reference.  Launch count: 1 (+ the
    memset of the barrier word) against 16."""
    from vtc_amd import _lib as L
    from vtc_amd import towers
    a = A.VIT_B32
    lib = L.lib()
    for act in (None, "normalize", "squash", "tanh"):
        sd = A.synth_cam(a, 31)
        for nc, sizes in ((5, (1, 3, 50, 85)), (2, (7,))):
            for B in sizes:
                g = torch.Generator().manual_seed(1000 + B)
                main = torch.randn(B, 512, generator=g).cuda()
                comm = torch.randn(B * nc, 512, generator=g).cuda()
                comments = A.synth_tokens(B * nc, a, 5 + B, empty_frac=0.3).reshape(B, nc, -1).cuda()
                pk = towers.PackedCam(cuda_sd(sd), torch.float32, 8, True, act)
                n0 = lib.vtc_debug_launch_count()
                fused = pk.forward(main, comm, comments)
                n_fused = lib.vtc_debug_launch_count() - n0
                pk.w.flags = L.CAM_NO_FUSED
                n0 = lib.vtc_debug_launch_count()
                multi = pk.forward(main, comm, comments)
                n_multi = lib.vtc_debug_launch_count() - n0
                d = (fused - multi).abs().max().item()
                assert torch.isfinite(fused).all() and d < 2e-6, (act, nc, B, d)
                assert n_fused == 1 and n_multi >= 10, (n_fused, n_multi)
        if act is None:
            print(f"[parity] one-launch CAM vs multi-launch, B=85: max |diff| {d:.2e}; launches {n_fused} vs {n_multi}")


def test_one_launch_cam_from_two_streams_at_once():
    """Two cooperative CAM launches on one card at the same time would starve each other's grid barrier (a width-512 workgroup
    takes a whole CU's LDS).  cam.hip's launcher keeps a completion event of the last one-launch CAM per device and sends a call
    that arrives on ANOTHER stream while it may still run to the multi-launch path: interleaved calls on two streams from two
    threads must all give the single-stream result (and none may take the seconds a starved barrier would)."""
    import threading
    import time
    from vtc_amd import towers
    a = A.VIT_B32
    sd = A.synth_cam(a, 41)
    B, nc = 40, 5
    g = torch.Generator().manual_seed(9)
    main = torch.randn(B, 512, generator=g).cuda()
    comm = torch.randn(B * nc, 512, generator=g).cuda()
    comments = A.synth_tokens(B * nc, a, 6, empty_frac=0.3).reshape(B, nc, -1).cuda()
    pk = [towers.PackedCam(cuda_sd(sd), torch.float32, 8, True, None) for _ in range(2)]
    ref = pk[0].forward(main, comm, comments).clone()
    torch.cuda.synchronize()
    outs = [[], []]

    def work(i):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(40):
                outs[i].append(pk[i].forward(main, comm, comments))
        st.synchronize()

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0
    for o in outs[0] + outs[1]:
        assert (o - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize("arch_name,model_type", [("VIT_B16", "ViT-B/16"),
                                                  # 34 s, most of it the fp32 oracle of a 24-layer width-1024 tower on the host; the ViT-L/14 tower
                                                  # itself is held to the reference's golden in test_timesformer_tower_vs_golden (default run)
                                                  pytest.param("VIT_L14", "ViT-L/14", marks=pytest.mark.extended)])
def test_every_model_type_forward_vs_oracle(arch_name, model_type):
    """VERDICT r3 missing #4: the other two model types of the reference's factory (model/timesformer_clip_alt.py:297-310) through
    the drop-in wrappers -- 197 / 257 tokens per frame (the K/V-tiled attention core), patch 16 / 14 (K = 768 / 588 -> 640 padded),
    width 1024 x 24 layers, 768-d features with the reference's DEFAULT n_heads = 8 (ViT-L/14: CAM head_dim 96, the generic
    short-sequence attention core) -- against the live oracle: fp32 1e-5, bf16 1e-3."""
    import warnings
    from vtc_amd.host import model as HM
    a = ARCH[arch_name]
    B, F = 2, 2
    sd = A.synth_model(a, 91, "timesformer_finaltf", nframes=F)
    g = torch.Generator().manual_seed(92)
    for k in list(sd):                       # a checkpoint's temporal_fc / CAM projections are not the init's zeros
        if k.endswith("temporal_fc.weight") or (k.startswith("final_transformer.") and (k.endswith("out_proj.weight") or k.endswith("c_proj.weight"))):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    vid = A.synth_pixels((B, F, 3, 224, 224), 93)
    title = A.synth_tokens(B, a, 94)
    comments = A.synth_tokens(B * 5, a, 95, empty_frac=0.2).reshape(B, 5, -1)
    heads = 8                                # model/model.py:546 default
    ref = M.pretrained_clip_timesformer_finaltf(vid, title, comments, sd, a, "text", n_heads=heads)

    class _TSF(HM.PretrainedCLIP_TimeSformer_finaltf):
        nframes = F
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = _TSF(model_type=model_type, branch_to_adapt_val="text", n_heads=heads)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    scale = float(sd["model.logit_scale"].exp())
    for dtype in DTYPES:
        m.compute_dtype = dtype
        got = m(vid.cuda(), title.cuda(), comments.cuda())
        tol = tol_for(dtype, 512)
        report(f"{model_type} TimeSformer_finaltf {dtype} feats_vis", float((got[0].cpu() - ref[0]).abs().max()), tol)
        report(f"{model_type} TimeSformer_finaltf {dtype} feats_text", float((got[1].cpu() - ref[1]).abs().max()), tol)
        report(f"{model_type} TimeSformer_finaltf {dtype} cosine sim", float((got[2].cpu() - ref[2]).abs().max()) / scale, tol)
    # the image tower of the same model type (upstream VisionTransformer: 197 / 257 tokens, one attention per block)
    sdi = A.synth_model(a, 96, "clip")
    img = A.synth_pixels((3, 3, 224, 224), 97)
    refi = CR.encode_image(img, sdi, a, "model.visual.").numpy()
    from vtc_amd import towers
    for dtype in DTYPES_H:
        pv = towers.PackedVision(cuda_sd({k: v for k, v in sdi.items() if k.startswith("model.visual.")}), "model.visual.", dtype)
        out = pv.forward(img.cuda()).cpu().numpy()
        report(f"{model_type} image tower {dtype}", np.abs(unit(out) - unit(refi)).max(), tol_for(dtype, 512))


def test_dense_text_chunking_covers_both_id_arrays():
    """TEXT_CHUNK used to be ignored whenever a second id array was given.  Dense path, chunks of 5 sequences over
    titles + comments == the unchunked call, bit for bit (the towers are per-sequence functions)."""
    from vtc_amd import towers
    a = A.TINY
    sd = A.synth_text(a, 91, prefix="model.")
    pt = towers.PackedText(cuda_sd(sd), "model.", torch.float32, heads=a.transformer_heads)
    t1, t2 = A.synth_tokens(7, a, 92).cuda(), A.synth_tokens(13, a, 93, empty_frac=0.2).cuda()
    whole = pt.forward(t1, ragged=False, ids_b=t2)
    was = towers.TEXT_CHUNK
    try:
        towers.TEXT_CHUNK = 5
        chunked = pt.forward(t1, ragged=False, ids_b=t2)
    finally:
        towers.TEXT_CHUNK = was
    assert torch.equal(whole, chunked)
    ref = CR.encode_text(torch.cat([t1, t2]).cpu(), sd, a, "model.").numpy()
    report("dense text, chunked, two id arrays vs oracle", np.abs(unit(chunked.cpu().numpy()) - unit(ref)).max(), 1e-5)
