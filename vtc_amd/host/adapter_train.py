"""Adapter-only training step on the HIP path.

What the reference does for ``configs/pretrained_clip_comments_attn_frozen.jsonc`` -- ``PretrainedCLIP_finaltf`` with
``freeze="all"`` in train mode, ``clip_loss``, ``loss.backward()``, ``torch.optim.Adam(lr=1e-3, amsgrad=True).step()``
(train.py:94-192, trainer/trainer.py, model/model.py:141-266) -- restricted to what is trainable there: the Context
Adapter Module (``final_transformer.*``) and ``mask_embedding``.  The frozen towers' outputs are inputs of the step
(run them with the forward path: ``vtc_amd.towers``).

Every arithmetic step is a HIP kernel behind the C ABI (``include/vtc_hip.h``, section "adapter-only training
step"): forward with saved activations from the existing primitives (vtc_gemm fp32, vtc_layernorm, vtc_attention,
vtc_normalize_rows, vtc_mean_groups, vtc_similarity, vtc_clip_loss), backward from ``vtc_*_bwd`` + vtc_gemm on
transposed operands, then ``vtc_adam_step``.  torch is used for allocation, views and copies only; there is no
autograd and no CPU fallback.  Oracle: ``oracle/train_ref.py`` (pinned by the reference's own train-mode run).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from .. import _lib as L
from .. import ops

SD = Dict[str, torch.Tensor]


def _lib():
    return L.lib()


def _st():
    return ops._stream()


def _t(x: torch.Tensor) -> torch.Tensor:                      # [r, c] -> [c, r]
    r, c = x.shape
    y = torch.empty(c, r, dtype=torch.float32, device=x.device)
    L.check(_lib().vtc_transpose_f32(x.data_ptr(), y.data_ptr(), r, c, _st()), "vtc_transpose_f32")
    return y


def _colsum(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    L.check(_lib().vtc_colsum_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], _st()), "vtc_colsum_f32")
    return out


def _axpby(x, y, a=1.0, b=1.0, out=None):
    out = torch.empty_like(x) if out is None else out
    L.check(_lib().vtc_axpby(out.data_ptr(), x.data_ptr(), y.data_ptr() if y is not None else None, a, b, x.numel(), _st()), "vtc_axpby")
    return out


def _scale_rows(x, s, group=1):
    L.check(_lib().vtc_scale_rows(x.data_ptr(), s.data_ptr(), x.shape[0], x.shape[1], group, _st()), "vtc_scale_rows")
    return x


def _norm_bwd(x, dy):
    dx = torch.empty_like(x)
    L.check(_lib().vtc_normalize_rows_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1], _st()), "vtc_normalize_rows_bwd")
    return dx


def _gelu(x, dy=None):
    out = torch.empty_like(x)
    L.check(_lib().vtc_quickgelu(x.data_ptr(), dy.data_ptr() if dy is not None else None, out.data_ptr(), x.numel(), _st()), "vtc_quickgelu")
    return out


def _ln_bwd(x, gamma, dy, dx_acc):
    dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    L.check(_lib().vtc_layernorm_bwd(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), dx_acc.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                     x.shape[0], x.shape[1], 1, _st()), "vtc_layernorm_bwd")
    return dg, db


def _linear_bwd(x, w, dy):
    """y = x w^T + b  ->  (dx = dy w, dw = dy^T x, db = colsum dy); x [R,I], w [O,I], dy [R,O]."""
    dx = ops.gemm(dy, _t(w), None)                             # [R,O] x ([I,O])^T
    dw = ops.gemm(_t(dy), _t(x), None)                         # [O,R] x ([I,R])^T
    return dx, dw, _colsum(dy)


class AdapterTrainer:
    """Holds the adapter parameters (fp32, updated in place) and the Adam state; ``step`` = forward + backward + update."""

    def __init__(self, sd: SD, n_layers: int = 2, n_heads: int = 8, branch: str = "text", lr: float = 1e-3,
                 betas=(0.9, 0.999), eps: float = 1e-8, amsgrad: bool = True):
        if branch not in ("text", "image"):
            raise Exception("Unknown branch_to_adapt")         # model/model.py:261 ("skip" trains nothing)
        self.n_layers, self.n_heads, self.branch = n_layers, n_heads, branch
        self.lr, self.betas, self.eps, self.amsgrad, self.t = lr, betas, eps, amsgrad, 0
        names = [k for k in sd if k.startswith("final_transformer.") or k == "mask_embedding"]
        self.params = {k: ops._gpu(sd[k], torch.float32, k).clone().contiguous() for k in names}
        self.logit_scale = ops._gpu(sd["model.logit_scale"], torch.float32, "logit_scale").reshape(1).clone()   # frozen ("all")
        z = lambda: {k: torch.zeros_like(v) for k, v in self.params.items()}
        self.m, self.v, self.vmax = z(), z(), z()
        self.grads: SD = {}

    def _p(self, l, name):
        return self.params[f"final_transformer.resblocks.{l}.{name}"]

    @ops.on_device
    def step(self, feats_vis: torch.Tensor, feats_title: torch.Tensor, feats_comm_raw: torch.Tensor, empty: torch.Tensor,
             skip_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feats_vis / feats_title [B,D], feats_comm_raw [nc,B,D] (text-tower outputs before the empty-comment
        substitution), empty [B,nc] bool, skip_mask [B] bool (the random_skip_adapter draw, model/model.py:199-201;
        True = adapter output zeroed) -> loss (0-d tensor on the GPU)."""
        fv_in = ops._gpu(feats_vis, torch.float32, "feats_vis")
        ft_in = ops._gpu(feats_title, torch.float32, "feats_title")
        fc = ops._gpu(feats_comm_raw, torch.float32, "feats_comm")
        dev = fv_in.device
        B, D = ft_in.shape
        nc = fc.shape[0]
        Lc = 1 + nc
        Bp = (B + 31) // 32 * 32                               # fp32 GEMM K granule (the batch is a K of dsim^T fv and of wgrad)
        R = Bp * Lc
        main = ft_in if self.branch == "text" else fv_in
        # ---- token matrix R0 [Bp, Lc, D]: main, comments (empty -> mask_embedding); pad items = e0 -----------
        R0 = torch.zeros(Bp, Lc, D, dtype=torch.float32, device=dev)
        R0[:, :, 0] = 1.0
        R0[:B, 0] = main
        R0[:B, 1:] = fc.permute(1, 0, 2)
        emp = torch.zeros(Bp, Lc, dtype=torch.bool, device=dev)
        emp[:B, 1:] = empty.to(dev)
        R0[emp] = self.params["mask_embedding"].reshape(-1)    # model/model.py:212
        R0 = R0.reshape(R, D)
        emp_rows = emp.reshape(R).to(torch.float32)
        x = ops.normalize_rows(R0)                             # :150-151
        saved = []
        for l in range(self.n_layers):
            x_in = x
            h1 = ops.layernorm(x_in, self._p(l, "ln_1.weight"), self._p(l, "ln_1.bias"))
            qkv = ops.gemm(h1, self._p(l, "attn.in_proj_weight"), self._p(l, "attn.in_proj_bias"))
            a = ops.attention(qkv, Bp, Lc, self.n_heads)
            x_mid = x_in.clone()
            ops.gemm(a, self._p(l, "attn.out_proj.weight"), self._p(l, "attn.out_proj.bias"), epilogue=L.EPI_RESID, out=x_mid)
            h2 = ops.layernorm(x_mid, self._p(l, "ln_2.weight"), self._p(l, "ln_2.bias"))
            pre = ops.gemm(h2, self._p(l, "mlp.c_fc.weight"), self._p(l, "mlp.c_fc.bias"))
            act = _gelu(pre)
            x = x_mid.clone()
            ops.gemm(act, self._p(l, "mlp.c_proj.weight"), self._p(l, "mlp.c_proj.bias"), epilogue=L.EPI_RESID, out=x)
            saved.append((x_in, h1, qkv, a, x_mid, h2, pre, act))
        Y = x
        Yn = ops.normalize_rows(Y)
        r0 = ops.mean_groups(Yn, Lc)                           # :157-159 mean over the 1 + nc tokens
        r = ops.normalize_rows(r0)
        keep = torch.ones(Bp, dtype=torch.float32, device=dev)
        if skip_mask is not None:
            keep[:B] = (~skip_mask.to(dev)).to(torch.float32)  # :199-201
        rm = _scale_rows(r.clone(), keep)
        main_p = torch.zeros(Bp, D, dtype=torch.float32, device=dev)
        main_p[:, 0] = 1.0
        main_p[:B] = main
        s = _axpby(ops.normalize_rows(main_p), rm)             # :203 normalize(main) + comm_res
        adapted = ops.normalize_rows(s)
        other = torch.zeros(Bp, D, dtype=torch.float32, device=dev)
        other[:, 0] = 1.0
        other[:B] = fv_in if self.branch == "text" else ft_in
        fa, fo = ops.normalize_rows(adapted), ops.normalize_rows(other)       # :263-264
        fvn, ftn = (fo, fa) if self.branch == "text" else (fa, fo)
        sim = ops.similarity(fvn[:B].contiguous(), ftn[:B].contiguous(), self.logit_scale)   # :478
        loss = ops.clip_loss(sim)
        # ---- backward -----------------------------------------------------------------------------------
        dsim = torch.empty_like(sim)
        ws = torch.empty(4 * B, dtype=torch.float32, device=dev)
        L.check(_lib().vtc_clip_loss_bwd(sim.data_ptr(), B, dsim.data_ptr(), ws.data_ptr(), ws.numel() * 4, _st()), "vtc_clip_loss_bwd")
        dsp = torch.zeros(Bp, Bp, dtype=torch.float32, device=dev)
        dsp[:B, :B] = dsim
        scale = float(self.logit_scale.exp().item())
        # sim = scale * fv ft^T : d ft = scale * dsim^T fv ; d fv = scale * dsim ft
        if self.branch == "text":
            dfa = ops.gemm(_t(dsp), _t(fvn), None)             # [Bp(text), Bp] x ([D, Bp])^T
        else:
            dfa = ops.gemm(dsp, _t(ftn), None)
        dfa = _axpby(dfa, None, scale, 0.0)
        dadapted = _norm_bwd(adapted, dfa)
        ds = _norm_bwd(s, dadapted)
        dr = _scale_rows(ds, keep)                             # d(normalize(main)) is not needed: main is frozen
        dr0 = _norm_bwd(r0, dr)
        dYn = _axpby(dr0.repeat_interleave(Lc, dim=0).contiguous(), None, 1.0 / Lc, 0.0)
        dx = _norm_bwd(Y, dYn)
        g: SD = {}
        for l in reversed(range(self.n_layers)):
            x_in, h1, qkv, a, x_mid, h2, pre, act = saved[l]
            pf = f"final_transformer.resblocks.{l}."
            dact, g[pf + "mlp.c_proj.weight"], g[pf + "mlp.c_proj.bias"] = _linear_bwd(act, self._p(l, "mlp.c_proj.weight"), dx)
            dpre = _gelu(pre, dact)
            dh2, g[pf + "mlp.c_fc.weight"], g[pf + "mlp.c_fc.bias"] = _linear_bwd(h2, self._p(l, "mlp.c_fc.weight"), dpre)
            dx_mid = dx.clone()
            g[pf + "ln_2.weight"], g[pf + "ln_2.bias"] = _ln_bwd(x_mid, self._p(l, "ln_2.weight"), dh2, dx_mid)
            da, g[pf + "attn.out_proj.weight"], g[pf + "attn.out_proj.bias"] = _linear_bwd(a, self._p(l, "attn.out_proj.weight"), dx_mid)
            dqkv = torch.empty_like(qkv)
            L.check(_lib().vtc_attention_small_bwd(qkv.data_ptr(), da.data_ptr(), dqkv.data_ptr(), Bp, Lc, self.n_heads, _st()),
                    "vtc_attention_small_bwd")
            dh1, g[pf + "attn.in_proj_weight"], g[pf + "attn.in_proj_bias"] = _linear_bwd(h1, self._p(l, "attn.in_proj_weight"), dqkv)
            dx_in = dx_mid.clone()
            g[pf + "ln_1.weight"], g[pf + "ln_1.bias"] = _ln_bwd(x_in, self._p(l, "ln_1.weight"), dh1, dx_in)
            dx = dx_in
        dR0 = _norm_bwd(R0, dx)
        g["mask_embedding"] = _colsum(_scale_rows(dR0, emp_rows)).reshape(1, D)
        self.grads = g
        # ---- Adam (amsgrad) -----------------------------------------------------------------------------
        self.t += 1
        for k, grad in g.items():
            p = self.params[k]
            L.check(_lib().vtc_adam_step(p.data_ptr(), grad.contiguous().data_ptr(), self.m[k].data_ptr(), self.v[k].data_ptr(),
                                         self.vmax[k].data_ptr(), p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                                         int(self.amsgrad), _st()), "vtc_adam_step")
        return loss
