"""CPU restatement of ONE adapter-only training step of the reference:
``PretrainedCLIP_finaltf`` with ``freeze="all"`` (configs/pretrained_clip_comments_attn_frozen.jsonc),
``clip_loss`` (model/loss.py:18-22), Adam with ``amsgrad=True`` (same config, ``optimizer``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned by tests/golden/train_step_*.npz, produced by the
reference's own classes in train mode + torch.optim.Adam (tests/golden/make_golden.py::gen_train_step).

With the towers frozen the trainable set is ``final_transformer.*``, ``final_linear.weight`` (no gradient when
``init_from_avg``) and ``mask_embedding`` (model/model.py:396-400; train.py:107 ``final_adapter_layers``); the towers'
outputs are constants of the step.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .arch import EOT
from .model_ref import RESIDUAL_ACTIVATIONS, clip_loss, normalize
from .clip_ref import transformer

SD = Dict[str, torch.Tensor]


def adapter_param_names(sd: SD):
    """train.py:107 ``final_adapter_layers`` matched against the wrapper's parameter names."""
    return [k for k in sd if k.startswith("final_transformer.") or k.startswith("final_linear.") or k == "mask_embedding"]


def train_forward(feats_vis: torch.Tensor, feats_title: torch.Tensor, feats_comm_raw: torch.Tensor, empty: torch.Tensor,
                  skip_mask: Optional[torch.Tensor], sd: SD, branch: str = "text", n_heads: int = 8,
                  init_from_avg: bool = True, residual_activation=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Train-mode ``forward`` downstream of the frozen towers (model/model.py:458-480 -> :216-266 -> :141-205).

    feats_vis / feats_title [B,D]: tower outputs; feats_comm_raw [nc,B,D]: text-tower outputs of the comments BEFORE
    the empty-comment substitution; empty [B,nc] bool (token 1 == EOT, :208); skip_mask [B] bool or None: the
    ``random_skip_adapter`` draw ``torch.rand(B) > 0.5`` (:199-201), True = adapter output zeroed.
    ``random_comment_masking`` is False in every config (masks of ones, :243-246: the multiplication is the identity).
    Returns (feats_vis_n, feats_text_n, sim)."""
    fc = feats_comm_raw.clone()
    fc[empty.t()] = sd["mask_embedding"].to(fc.dtype)                          # :212 (rows of a [nc,B,D] tensor)
    main = feats_title if branch == "text" else feats_vis
    x = normalize(torch.cat([main[None], fc], dim=0))                          # :150-151
    y = transformer(x.transpose(0, 1), sd, "final_transformer", n_heads).transpose(0, 1)   # :155
    if init_from_avg:
        r = normalize(torch.mean(normalize(y), dim=0))                         # :157-159
    else:
        r = y[0] @ sd["final_linear.weight"].t()                               # :161
    r = RESIDUAL_ACTIVATIONS[residual_activation](r)
    if skip_mask is not None:
        r = r * (~skip_mask)[:, None].to(r.dtype)                              # :199-201 comm_res[comm_mask] = 0.0
    adapted = normalize(normalize(main) + r)                                   # :203
    fv, ft = (feats_vis, adapted) if branch == "text" else (adapted, feats_title)
    fv, ft = normalize(fv), normalize(ft)                                      # :263-264
    sim = sd["model.logit_scale"].exp() * fv @ ft.t()                          # :478
    return fv, ft, sim


class AdamAmsgrad:
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True), restated per tensor
    (torch/optim/adam.py single-tensor path: bias-corrected step size, max of the raw second moments)."""

    def __init__(self, params: SD, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, betas[0], betas[1], eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.vmax = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params: SD, grads: SD):
        self.t += 1
        bc1, bc2 = 1 - self.b1 ** self.t, 1 - self.b2 ** self.t
        for k, g in grads.items():
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            torch.maximum(self.vmax[k], self.v[k], out=self.vmax[k])
            denom = (self.vmax[k].sqrt() / bc2 ** 0.5).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def train_step(feats_vis, feats_title, feats_comm_raw, empty, skip_mask, sd: SD, opt: AdamAmsgrad, **cam):
    """One step: loss, gradients of the adapter parameters (autograd over the restated forward), Adam update in
    place.  Returns (loss, grads)."""
    names = [k for k in adapter_param_names(sd)]
    work = dict(sd)
    leaves = {}
    for k in names:
        leaves[k] = sd[k].detach().clone().requires_grad_(True)
        work[k] = leaves[k]
    used = [k for k in names if not (k.startswith("final_linear.") and cam.get("init_from_avg", True))]
    with torch.enable_grad():
        _, _, sim = train_forward(feats_vis, feats_title, feats_comm_raw, empty, skip_mask, work, **cam)
        loss = clip_loss(sim)
        gs = torch.autograd.grad(loss, [leaves[k] for k in used])
    grads = {k: g for k, g in zip(used, gs)}
    with torch.no_grad():
        opt.step({k: sd[k] for k in used}, grads)
    return float(loss), grads
