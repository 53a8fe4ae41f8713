"""CPU restatement of model/model.py (wrappers + Context Adapter Module) and model/loss.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  First-party reference
arithmetic, pinned by tests/golden (the reference's own classes run unmodified).
Eval-mode semantics only: the train-only branches (random comment masking
model/model.py:236-246, random_skip_adapter :199-201) are out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .arch import ClipArch, EOT
from .clip_ref import SD, encode_image, encode_text, transformer
from .timesformer_ref import timesformer_alt


def normalize(x: torch.Tensor) -> torch.Tensor:
    """model/model.py:26-27."""
    return x / x.norm(dim=-1, keepdim=True)


def squash(s: torch.Tensor) -> torch.Tensor:
    """model/model.py:34-39."""
    s = s + 1e-9
    mag_sq = torch.sum(s ** 2, dim=-1, keepdim=True)
    mag = torch.sqrt(mag_sq)
    return (mag_sq / (1.0 + mag_sq)) * (s / mag)


# model/model.py:65-77; the two stateful entries ("sub_mean" :42-51, "bn" :54-61) are applied in adapt_feature from
# the mean_center_bn running statistics (eval semantics)
RESIDUAL_ACTIVATIONS = {
    "normalize": lambda x: normalize(x + 1e-9),
    "squash": squash,
    "squash10": lambda x: 10 * squash(x),
    "squash1p2": lambda x: 1.2 * squash(x),
    "squash1p5": lambda x: 1.5 * squash(x),
    "squash1p8": lambda x: 1.8 * squash(x),
    "tanh": torch.tanh,
    "none": lambda x: x,
    None: lambda x: x,
}


def load_comment_features(comments: torch.Tensor, sd: SD, arch: ClipArch) -> torch.Tensor:
    """_load_comment_features, model/model.py:207-214.
    comments [B,nc,ctx] int64 -> [nc,B,D]; rows whose token 1 is EOT (empty string)
    are overwritten with mask_embedding (:208,:212)."""
    empty = comments[..., 1] == EOT
    b, nc, nt = comments.shape
    f = encode_text(comments.reshape(b * nc, nt), sd, arch, "model.").reshape(b, nc, -1).float()
    f = f.clone()
    f[empty] = sd["mask_embedding"].to(f.dtype)
    return f.permute(1, 0, 2)


def adapt_feature(main: torch.Tensor, aux: torch.Tensor, sd: SD, n_heads: int = 8,
                  init_from_avg: bool = True, residual_activation=None) -> torch.Tensor:
    """_adapt_feature, model/model.py:141-205 (eval).  main [B,D], aux [nc,B,D]."""
    x = normalize(torch.cat([main[None], aux], dim=0))             # :150-151  [1+nc, B, D]
    # final_transformer is a clip.model.Transformer applied sequence-first (:155): each
    # batch item's 1+nc tokens form one unmasked sequence.
    y = transformer(x.transpose(0, 1), sd, "final_transformer", n_heads).transpose(0, 1)
    if init_from_avg:
        r = normalize(torch.mean(normalize(y), dim=0))              # :157-159
    else:
        r = y[0] @ sd["final_linear.weight"].t()                    # :161
    if residual_activation == "sub_mean":                          # :50  s - running_mean
        r = r - sd["mean_center_bn.running_mean"]
    elif residual_activation == "bn":                              # :60  BatchNorm1d(affine=False) in eval mode
        r = (r - sd["mean_center_bn.running_mean"]) / torch.sqrt(sd["mean_center_bn.running_var"] + 1e-5)
    else:
        r = RESIDUAL_ACTIVATIONS[residual_activation](r)           # :168-171
    return normalize(normalize(main) + r)                           # :203


def encode_with_comments(fv, ft, comments, sd: SD, arch: ClipArch, branch: str, **cam):
    """_encode_with_comments (eval path), model/model.py:216-266."""
    fc = load_comment_features(comments, sd, arch)
    if branch == "text":
        ft = adapt_feature(ft, fc, sd, **cam)
    elif branch == "image":
        fv = adapt_feature(fv, fc, sd, **cam)
    elif branch != "skip":
        raise Exception("Unknown branch_to_adapt")                  # :261
    return normalize(fv), normalize(ft)                             # :263-264


def _sim(fv, ft, sd):
    return sd["model.logit_scale"].exp() * fv @ ft.t()              # :369,478,504,621


def _encode_vis_clip(vis, sd, arch):
    """vis.ndim dispatch of PretrainedCLIP*.forward (model/model.py:327-338, :459-470)."""
    if vis.ndim == 2:
        return vis
    if vis.ndim == 4:
        return encode_image(vis, sd, arch, "model.visual.").float()
    s = vis.shape
    f = encode_image(vis.reshape(s[0] * s[1], *s[2:]), sd, arch, "model.visual.").float()
    return f.reshape(s[0], s[1], -1).mean(1)


def pretrained_clip(vis, title, sd: SD, arch: ClipArch, comments=None, comment_fusion=None):
    """PretrainedCLIP.forward, model/model.py:326-371."""
    fv = _encode_vis_clip(vis, sd, arch)
    ft = encode_text(title, sd, arch, "model.")
    if not (comments is None or comment_fusion is None or comment_fusion == "None"):
        if comment_fusion != "averaging":
            raise Exception("Comment fusion method not specified.")  # :364
        b, nc, nt = comments.shape
        fc = encode_text(comments.reshape(b * nc, nt), sd, arch, "model.").reshape(b, nc, -1).float()
        ft = torch.mean(torch.cat([ft[None], fc.permute(1, 0, 2)], 0), dim=0)   # :357-362
    ft, fv = normalize(ft), normalize(fv)
    return fv, ft, _sim(fv, ft, sd)


def pretrained_clip_finaltf(vis, title, comments, sd: SD, arch: ClipArch, branch="text", **cam):
    """PretrainedCLIP_finaltf.forward, model/model.py:458-480."""
    fv = _encode_vis_clip(vis, sd, arch)
    ft = encode_text(title, sd, arch, "model.")
    fv, ft = encode_with_comments(fv, ft, comments, sd, arch, branch, **cam)
    return fv, ft, _sim(fv, ft, sd)


def pretrained_clip_timesformer(im, text, sd: SD, arch: ClipArch):
    """PretrainedCLIP_TimeSformer.forward, model/model.py:494-506."""
    fv = normalize(timesformer_alt(im, sd, arch, "model.visual."))
    ft = normalize(encode_text(text, sd, arch, "model."))
    return fv, ft, _sim(fv, ft, sd)


def pretrained_clip_timesformer_finaltf(vis, title, comments, sd: SD, arch: ClipArch, branch="text", **cam):
    """PretrainedCLIP_TimeSformer_finaltf.forward, model/model.py:596-623."""
    fv = timesformer_alt(vis, sd, arch, "model.visual.")
    ft = encode_text(title, sd, arch, "model.")
    fv, ft = encode_with_comments(fv.float(), ft.float(), comments, sd, arch, branch, **cam)
    return fv, ft, _sim(fv, ft, sd)


def clip_loss(sim: torch.Tensor) -> torch.Tensor:
    """model/loss.py:18-22."""
    labels = torch.arange(sim.shape[0])
    return 0.5 * (F.cross_entropy(sim, labels) + F.cross_entropy(sim.t(), labels))
