"""CPU: oracle/ (the restatement) vs golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load_golden
from oracle import arch as A
from oracle import model_ref as M
from oracle import timesformer_ref as T

torch.set_grad_enabled(False)
ARCH = {"TINY": A.TINY, "VIT_B32": A.VIT_B32, "VIT_B16": A.VIT_B16, "VIT_L14": A.VIT_L14}
TOL = dict(rtol=0, atol=2e-5)  # fp32 CPU, different op order (einops/MHA vs index-wise)


@pytest.mark.parametrize("fname", golden_files("tower_"))
def test_visual_tower(fname):
    case, g = load_golden(fname)
    a = ARCH[case["arch"]]
    if case["arch"] == "VIT_B32" and case["nframes"] == 16:
        pytest.skip("covered on the GPU box; keeps the CPU suite short") if False else None
    sd = A.synth_visual(a, case["wseed"], nframes=case["nframes"], variant=case["variant"])
    x = A.synth_pixels((case["B"], case["nframes"], 3, a.image_resolution, a.image_resolution), case["xseed"])
    fn = T.timesformer_alt if case["variant"] == "alt" else T.timesformer_v1
    out = fn(x, sd, a, p="").numpy()
    np.testing.assert_allclose(out, g["out"], **TOL)


def run_wrapper(case):
    a = ARCH[case["arch"]]
    kind = case["model"]
    sd = A.synth_model(a, case["wseed"], kind, nframes=8, bn_stats=case["ctor"].get("residual_activation") in ("sub_mean", "bn"))
    vis = A.synth_pixels(case["vis_shape"], case["xseed"])
    title = A.synth_tokens(case["B"], a, case["tseed"])
    comments = A.synth_tokens(case["B"] * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(case["B"], 5, -1)
    kw = dict(case["ctor"])
    kw.pop("freeze", None)
    branch = kw.pop("branch_to_adapt_val", "text")
    cam = {}
    if "init_from_avg" in kw:
        cam["init_from_avg"] = kw.pop("init_from_avg")
    if "residual_activation" in kw:
        cam["residual_activation"] = kw.pop("residual_activation")
    if "n_heads" in kw:
        cam["n_heads"] = kw.pop("n_heads")
    if kind == "clip":
        return M.pretrained_clip(vis, title, sd, a, comments if case["comments"] else None, kw.get("comment_fusion"))
    if kind == "clip_finaltf":
        return M.pretrained_clip_finaltf(vis, title, comments, sd, a, branch, **cam)
    if kind == "timesformer":
        return M.pretrained_clip_timesformer(vis, title, sd, a)
    return M.pretrained_clip_timesformer_finaltf(vis, title, comments, sd, a, branch, **cam)


@pytest.mark.parametrize("fname", golden_files("wrap_"))
def test_wrappers(fname):
    case, g = load_golden(fname)
    fv, ft, sim = run_wrapper(case)
    np.testing.assert_allclose(fv.numpy(), g["feats_vis"], **TOL)
    np.testing.assert_allclose(ft.numpy(), g["feats_text"], **TOL)
    np.testing.assert_allclose(sim.numpy(), g["sim"], rtol=0, atol=2e-4)  # scaled by exp(logit_scale) ~ 14.3


def test_cam_at_init_closed_form():
    """model.py:440-450: with the init_from_avg zeroing the CAM transformer is an identity, so
    adapted = normalize(normalize(main) + normalize(mean_i normalize(token_i)))  (SURVEY 4)."""
    case, g = load_golden("cam_at_init_tiny.npz")
    a = A.TINY
    sd = A.synth_model(a, case["wseed"], "clip_finaltf", cam_at_init=True)
    B = case["B"]
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), case["xseed"])
    title = A.synth_tokens(B, a, case["tseed"])
    comments = A.synth_tokens(B * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(B, 5, -1)
    fv, ft, sim = M.pretrained_clip_finaltf(vis, title, comments, sd, a, "text", n_heads=2)
    np.testing.assert_allclose(ft.numpy(), g["feats_text"], **TOL)
    np.testing.assert_allclose(fv.numpy(), g["feats_vis"], **TOL)
    # closed form
    from oracle.clip_ref import encode_text
    main = encode_text(title, sd, a, "model.")
    fc = M.load_comment_features(comments, sd, a)
    toks = M.normalize(torch.cat([main[None], fc], 0))
    closed = M.normalize(M.normalize(main) + M.normalize(toks.mean(0)))
    np.testing.assert_allclose(M.normalize(closed).numpy(), g["feats_text"], **TOL)
    # unit norm outputs (model.py:263-264)
    np.testing.assert_allclose(np.linalg.norm(g["feats_text"], axis=-1), 1.0, atol=1e-6)


def test_clip_loss_golden():
    case, g = load_golden("clip_loss.npz")
    for i in range(3):
        v = float(M.clip_loss(torch.from_numpy(g[f"sim{i}"])))
        assert abs(v - g["loss"][i]) < 1e-6
    # analytic: sim = c*I  ->  loss = log(1 + (n-1) e^-c)
    n, c = 6, 3.0
    v = float(M.clip_loss(torch.eye(n) * c))
    assert abs(v - np.log(1 + (n - 1) * np.exp(-c))) < 1e-6


def test_identity_at_init_timesformer_equals_vit():
    """SURVEY 4 known answer: temporal_fc = 0 and F identical frames => TimeSformer(alt) == ViT."""
    from oracle.clip_ref import encode_image
    a = A.TINY
    sd = A.synth_visual(a, 5, nframes=4)
    for k in list(sd):
        if "temporal_fc" in k:
            sd[k] = torch.zeros_like(sd[k])
    sd["temporal_embed"] = torch.zeros_like(sd["temporal_embed"])
    img = A.synth_pixels((2, 1, 3, a.image_resolution, a.image_resolution), 6)
    vid = img.expand(2, 4, 3, a.image_resolution, a.image_resolution).contiguous()
    tf = T.timesformer_alt(vid, sd, a, p="")
    vit = encode_image(img[:, 0], sd, a, p="")
    np.testing.assert_allclose(tf.numpy(), vit.numpy(), atol=5e-6)


def test_train_step_oracle_vs_reference_golden():
    """Adapter-only training step: the oracle's restated train-mode forward + autograd + restated
    Adam(amsgrad) reproduce the reference's loss, every adapter gradient and the parameters after two steps."""
    from dataclasses import asdict  # noqa: F401
    from oracle import clip_ref as CR
    from oracle import train_ref as TR
    case, g = load_golden("train_step_tiny.npz")
    a = A.TINY
    sd = A.synth_model(a, case["wseed"], "clip_finaltf")
    B = case["B"]
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), case["xseed"])
    title = A.synth_tokens(B, a, case["tseed"])
    comments = A.synth_tokens(B * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(B, 5, -1)
    with torch.no_grad():                                   # frozen towers: constants of the step
        fv = CR.encode_image(vis, sd, a, "model.visual.").float()
        ft = CR.encode_text(title, sd, a, "model.").float()
        fc = CR.encode_text(comments.reshape(B * 5, -1), sd, a, "model.").float().reshape(B, 5, -1).permute(1, 0, 2)
    empty = comments[..., 1] == A.EOT
    params = {k: sd[k].clone() for k in TR.adapter_param_names(sd)}
    sd = dict(sd); sd.update(params)
    opt = TR.AdamAmsgrad({k: v for k, v in params.items()}, lr=case["lr"])
    for step, seed in enumerate(case["rng_seeds"]):
        torch.manual_seed(seed)
        torch.rand([])                                       # model/model.py:163 consumes one draw first
        skip = torch.rand(B) > 0.5                           # :200
        loss, grads = TR.train_step(fv, ft, fc, empty, skip, sd, opt, n_heads=case["n_heads"])
        assert abs(loss - float(g[f"loss{step}"])) < 2e-6 * max(1.0, abs(loss)), (step, loss, float(g[f"loss{step}"]))
        if step == 0:
            for k, v in grads.items():
                ref = g["grad0:" + k]
                assert np.abs(v.numpy() - ref).max() < 2e-6 * max(1e-3, np.abs(ref).max()) + 1e-8, k
    for k in params:
        d = np.abs(sd[k].numpy() - g["after2:" + k])
        # Adam's step is lr * m / (sqrt(v) + 1e-8): where |grad| ~ 1e-8 its direction is decided by rounding noise, so
        # elements with a vanishing gradient are only bounded by the two steps' maximum travel (2 lr)
        big = np.abs(g["grad0:" + k]) > 1e-5 if ("grad0:" + k) in g else np.zeros(d.shape, bool)
        assert d[big].max(initial=0.0) < 5e-6, k
        assert d.max() <= 2.001 * case["lr"], k
