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
ARCH = {"TINY": A.TINY, "VIT_B32": A.VIT_B32}
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
    sd = A.synth_model(a, case["wseed"], kind, nframes=8)
    vis = A.synth_pixels(case["vis_shape"], case["xseed"])
    title = A.synth_tokens(case["B"], a, case["tseed"])
    comments = A.synth_tokens(case["B"] * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(case["B"], 5, -1)
    kw = dict(case["ctor"])
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
