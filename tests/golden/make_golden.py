"""Generate tests/golden/*.npz by running the REFERENCE's own Python (unmodified).

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_golden.py [/root/reference]

The reference's first-party arithmetic (model/timesformer_clip_alt.py,
model/timesformer_clip.py, model/model.py, model/loss.py) is imported as-is; the
un-vendored third-party ``clip`` package is replaced by tests/golden/clip_double.py.
Weights and inputs are regenerated from seeds by oracle/arch.py, so each fixture
stores only the case description (as JSON) and the expected outputs.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import clip_double  # noqa: E402

clip_double.install()
sys.path.insert(0, REF)  # reference's ``model`` package shadows nothing of ours in this process

from oracle import arch as A  # noqa: E402

import model.loss as ref_loss  # noqa: E402  (reference)
import model.model as ref_model  # noqa: E402  (reference)
import model.timesformer_clip as ref_tf1  # noqa: E402  (reference)
import model.timesformer_clip_alt as ref_alt  # noqa: E402  (reference)

assert ref_model.__file__.startswith(REF), ref_model.__file__
torch.manual_seed(0)
torch.set_grad_enabled(False)


def save(name, case, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, case=np.array(json.dumps(case)), **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote", os.path.relpath(path, ROOT), {k: tuple(np.asarray(v).shape) for k, v in arrays.items()})


def arch_of(name):
    return {"TINY": A.TINY, "VIT_B32": A.VIT_B32, "VIT_B16": A.VIT_B16, "VIT_L14": A.VIT_L14}[name]


def visual_tower(variant, arch_name, nframes, B, wseed, xseed):
    a = arch_of(arch_name)
    cls = ref_alt.VisualTransformer if variant == "alt" else ref_tf1.VisualTransformer
    m = cls(a.image_resolution, a.vision_patch_size, a.vision_width, a.vision_layers, a.vision_heads,
            a.embed_dim, nframes).eval()
    m.load_state_dict(A.synth_visual(a, wseed, nframes=nframes, variant=variant), strict=True)
    x = A.synth_pixels((B, nframes, 3, a.image_resolution, a.image_resolution), xseed)
    return m(x).numpy()


def gen_towers():
    for variant in ("alt", "v1"):
        for arch_name, F_, B in (("TINY", 1, 2), ("TINY", 2, 2), ("TINY", 8, 2), ("VIT_B32", 8, 1)) + \
                ((("VIT_B32", 16, 1),) if variant == "alt" else ()):
            case = dict(kind="visual_tower", variant=variant, arch=arch_name, nframes=F_, B=B, wseed=11, xseed=12)
            save(f"tower_{variant}_{arch_name.lower()}_f{F_}", case,
                 out=visual_tower(variant, arch_name, F_, B, 11, 12))


def gen_towers_b16_l14():
    """The other two model types of make_timesformer_clip_vit_alt (model/timesformer_clip_alt.py:297-310): 197 and 257 tokens per
    frame (space attention beyond 80 keys), patch 16 and 14 (K = 768 / 588), width 1024 x 24 layers."""
    for arch_name in ("VIT_B16", "VIT_L14"):
        case = dict(kind="visual_tower", variant="alt", arch=arch_name, nframes=2, B=1, wseed=11, xseed=12)
        save(f"tower_alt_{arch_name.lower()}_f2", case, out=visual_tower("alt", arch_name, 2, 1, 11, 12))


def build_wrapper(kind, arch_name, seed, **kw):
    """Construct the reference wrapper unmodified, then load the seeded state dict strictly."""
    a = arch_of(arch_name)
    clip_double.ARCH = a
    cls = {"clip": ref_model.PretrainedCLIP, "clip_finaltf": ref_model.PretrainedCLIP_finaltf,
           "timesformer": ref_model.PretrainedCLIP_TimeSformer,
           "timesformer_finaltf": ref_model.PretrainedCLIP_TimeSformer_finaltf}[kind]
    m = cls(**kw).eval()
    sd = A.synth_model(a, seed, kind, nframes=8, bn_stats=kw.get("residual_activation") in ("sub_mean", "bn"))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # the double's causal mask etc. are not parameters; every parameter must be covered
    assert not unexpected, unexpected
    assert not missing, missing
    return m, a


def gen_wrappers():
    torch.manual_seed(0)
    # (kind, arch, B, vis ndim, ctor kwargs)
    cases = [
        ("clip", "TINY", 3, 4, {}),
        ("clip", "TINY", 2, 5, {}),                                  # frames -> mean over time (model.py:333-338)
        ("clip", "TINY", 3, 4, {"comment_fusion": "averaging"}),
        # TINY embed_dim is 128: n_heads=2 keeps the CAM head_dim at 64 as in the real model (512/8)
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "image", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "skip", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "init_from_avg": False, "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "residual_activation": "squash", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "residual_activation": "tanh", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "image", "residual_activation": "normalize", "n_heads": 2}),
        ("clip_finaltf", "VIT_B32", 2, 4, {"branch_to_adapt_val": "text"}),
        ("timesformer", "VIT_B32", 2, 5, {}),
        ("timesformer_finaltf", "VIT_B32", 2, 5, {"branch_to_adapt_val": "text"}),
        ("timesformer_finaltf", "VIT_B32", 2, 5, {"branch_to_adapt_val": "image"}),
        # stateful residual activations (model.py:42-61), eval semantics = the mean_center_bn running statistics;
        # "bn" reads state.branch_to_freeze with `in`, so the reference needs a string there (freeze="none")
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "residual_activation": "sub_mean", "n_heads": 2}),
        ("clip_finaltf", "TINY", 4, 4, {"branch_to_adapt_val": "text", "residual_activation": "bn", "freeze": "none", "n_heads": 2}),
    ]
    for i, (kind, arch_name, B, nd, kw) in enumerate(cases):
        m, a = build_wrapper(kind, arch_name, seed=21, **kw)
        res = a.image_resolution
        shape = (B, 3, res, res) if nd == 4 else (B, 8 if kind.startswith("timesformer") else 3, 3, res, res)
        vis = A.synth_pixels(shape, 22)
        title = A.synth_tokens(B, a, 23)
        comments = A.synth_tokens(B * 5, a, 24, empty_frac=0.3).reshape(B, 5, -1)
        needs_comments = kind.endswith("finaltf") or kw.get("comment_fusion")
        out = m(vis, title, comments) if needs_comments else m(vis, title)
        case = dict(kind="wrapper", model=kind, arch=arch_name, B=B, vis_shape=list(shape), ctor=kw, wseed=21,
                    xseed=22, tseed=23, cseed=24, empty_frac=0.3, comments=bool(needs_comments))
        tag = "_".join(f"{k[:6]}-{v}" for k, v in kw.items()) or "default"
        save(f"wrap_{i:02d}_{kind}_{arch_name.lower()}_{tag}", case,
             feats_vis=out[0].numpy(), feats_text=out[1].numpy(), sim=out[2].numpy())


def gen_cam_at_init():
    """The state tests/test_pretrained_clip.py:45-85 pins: init_from_avg zeroing makes the CAM
    transformer an identity (model.py:440-450)."""
    m, a = build_wrapper("clip_finaltf", "TINY", seed=31, branch_to_adapt_val="text", n_heads=2)
    sd = A.synth_model(a, 31, "clip_finaltf", cam_at_init=True)
    m.load_state_dict(sd, strict=True)
    B = 3
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), 32)
    title = A.synth_tokens(B, a, 33)
    comments = A.synth_tokens(B * 5, a, 34, empty_frac=0.3).reshape(B, 5, -1)
    out = m(vis, title, comments)
    save("cam_at_init_tiny", dict(kind="cam_at_init", arch="TINY", B=B, wseed=31, xseed=32, tseed=33, cseed=34,
                                  empty_frac=0.3),
         feats_vis=out[0].numpy(), feats_text=out[1].numpy(), sim=out[2].numpy())


def gen_loss():
    sims, vals = [], []
    for seed, n in ((0, 4), (1, 7), (2, 50)):
        torch.manual_seed(seed)
        s = torch.randn(n, n) * (3.0 if seed else 1.0)
        vals.append(float(ref_loss.clip_loss((None, None, s), None)))
        sims.append(s.numpy())
    save("clip_loss", dict(kind="clip_loss", n=[4, 7, 50]),
         sim0=sims[0], sim1=sims[1], sim2=sims[2], loss=np.array(vals, dtype=np.float64))


def gen_train_step():
    """One (and a second) adapter-only training step of the reference: PretrainedCLIP_finaltf(freeze="all") in
    train mode, clip_loss, torch.optim.Adam(lr=1e-3, amsgrad=True) over the parameters train.py:107 calls
    final_adapter_layers (configs/pretrained_clip_comments_attn_frozen.jsonc).  The random_skip_adapter draw
    (model.py:199-201) is made reproducible by seeding torch's generator right before each forward."""
    m, a = build_wrapper("clip_finaltf", "TINY", seed=41, freeze="all", branch_to_adapt="text", branch_to_adapt_val="text",
                         n_heads=2)
    m.train()
    torch.set_grad_enabled(True)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert all(n.startswith(("final_transformer.", "final_linear.")) or n == "mask_embedding" for n in names), names
    opt = torch.optim.Adam([p for _, p in m.named_parameters() if p.requires_grad], lr=1e-3, weight_decay=0, amsgrad=True)
    B = 8
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), 42)
    title = A.synth_tokens(B, a, 43)
    comments = A.synth_tokens(B * 5, a, 44, empty_frac=0.3).reshape(B, 5, -1)
    out = {}
    for step in range(2):
        torch.manual_seed(100 + step)                  # consumed by :163 torch.rand([]) then :200 torch.rand(B)
        opt.zero_grad()
        res = m(vis, title, comments)
        loss = ref_loss.clip_loss(res, None)
        loss.backward()
        out[f"loss{step}"] = float(loss)
        if step == 0:
            for n, p in m.named_parameters():
                if p.requires_grad and p.grad is not None:
                    out["grad0:" + n] = p.grad.detach().numpy().copy()
        opt.step()
    for n, p in m.named_parameters():
        if p.requires_grad:
            out["after2:" + n] = p.detach().numpy().copy()
    torch.set_grad_enabled(False)
    save("train_step_tiny", dict(kind="train_step", arch="TINY", B=B, wseed=41, xseed=42, tseed=43, cseed=44, empty_frac=0.3,
                                 rng_seeds=[100, 101], n_heads=2, lr=1e-3), **out)


if __name__ == "__main__":
    if os.environ.get("GOLDEN_ONLY"):
        globals()[os.environ["GOLDEN_ONLY"]]()
        sys.exit(0)
    gen_loss()
    gen_towers()
    gen_towers_b16_l14()
    gen_cam_at_init()
    gen_wrappers()
    gen_train_step()
