"""CPU restatement of the reference's two TimeSformer video towers.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  First-party reference
arithmetic: pinned by golden vectors produced by the reference's own classes
(tests/golden/make_golden.py).

Written index-wise (explicit gather/scatter of token rows) rather than with the
reference's einops rearranges, so that it documents the token order the HIP
kernels address directly.
"""
from __future__ import annotations

import torch

from .arch import ClipArch
from .clip_ref import SD, layer_norm, mha, mlp, n_layers, patch_embed


# --------------------------------------------------------------------------------------
# model/timesformer_clip_alt.py  (the variant model/model.py:11,488,557 actually uses)
# --------------------------------------------------------------------------------------

def alt_embed(video: torch.Tensor, sd: SD, p: str = "") -> torch.Tensor:
    """VisualTransformer.forward lines :253-277.

    video [B,F,3,H,W] -> tokens [B, 1 + P*F, W] ordered (cls, then patch-major /
    time-minor: row 1 + n*F + t is patch n of frame t, :271-274), +pos (:266),
    +temporal_embed[t] on patches (:272), cls = class_embedding + pos[0] (:269 keeps the
    first B cls rows, all identical), then ln_pre (:277)."""
    B, Fr = video.shape[:2]
    x = patch_embed(video.reshape(B * Fr, *video.shape[2:]), sd, p)          # [(b t), P, W]
    P, W = x.shape[1], x.shape[2]
    pos = sd[f"{p}positional_embedding"].to(x.dtype)
    x = x + pos[1:][None]                                                     # :266 (patch rows)
    x = x.reshape(B, Fr, P, W) + sd[f"{p}temporal_embed"].to(x.dtype)[None, :Fr, None, :]  # :271-272
    x = x.permute(0, 2, 1, 3).reshape(B, P * Fr, W)                           # (n t) order, :274
    cls = (sd[f"{p}class_embedding"].to(x.dtype) + pos[0]).expand(B, 1, W)
    x = torch.cat([cls, x], dim=1)                                            # :275
    return layer_norm(x, sd[f"{p}ln_pre.weight"], sd[f"{p}ln_pre.bias"])


def alt_block(x: torch.Tensor, sd: SD, p: str, heads: int, B: int, Fr: int) -> torch.Tensor:
    """ResidualAttentionBlock.forward, model/timesformer_clip_alt.py:135-175."""
    T, W = x.shape[1], x.shape[2]
    P = (T - 1) // Fr
    # temporal (:142-149): sequences = the F rows of one (video, patch); cls excluded
    xt = x[:, 1:, :].reshape(B * P, Fr, W)
    rt = mha(layer_norm(xt, sd[f"{p}.ln_time.weight"], sd[f"{p}.ln_time.bias"]), sd, f"{p}.timeattn", heads)
    rt = rt.reshape(B, P * Fr, W) @ sd[f"{p}.temporal_fc.weight"].t() + sd[f"{p}.temporal_fc.bias"]
    xt = x[:, 1:, :] + rt
    # spatial (:152-158): sequences = (video, frame): [cls, the P patches of that frame]
    cls0 = x[:, 0:1, :]
    xs = xt.reshape(B, P, Fr, W).permute(0, 2, 1, 3).reshape(B * Fr, P, W)
    cls_rep = cls0.expand(B, Fr, W).reshape(B * Fr, 1, W)
    xs = torch.cat([cls_rep, xs], dim=1)
    rs = mha(layer_norm(xs, sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"]), sd, f"{p}.attn", heads)
    # cls = mean over frames of the per-frame cls outputs (:162-164)
    cls_out = rs[:, 0, :].reshape(B, Fr, W).mean(1, keepdim=True)
    rs = rs[:, 1:, :].reshape(B, Fr, P, W).permute(0, 2, 1, 3).reshape(B, P * Fr, W)  # back to (n t), :166-168
    x = torch.cat([cls0, xt], dim=1) + torch.cat([cls_out, rs], dim=1)        # :173
    return x + mlp(layer_norm(x, sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"]), sd, f"{p}.mlp")  # :174


def timesformer_alt(video: torch.Tensor, sd: SD, arch: ClipArch, p: str = "visual.") -> torch.Tensor:
    """model/timesformer_clip_alt.py:252-286.  video [B,F,3,H,W] -> [B, embed_dim]."""
    B, Fr = video.shape[:2]
    x = alt_embed(video, sd, p)
    for i in range(n_layers(sd, f"{p}transformer")):
        x = alt_block(x, sd, f"{p}transformer.resblocks.{i}", arch.vision_heads, B, Fr)
    x = layer_norm(x[:, 0, :], sd[f"{p}ln_post.weight"], sd[f"{p}ln_post.bias"])   # :281
    return x @ sd[f"{p}proj"]                                                       # :284


# --------------------------------------------------------------------------------------
# model/timesformer_clip.py  (older variant; exported, not used by model/model.py)
# --------------------------------------------------------------------------------------

def _v1_masked_attention(x: torch.Tensor, sd: SD, p: str, heads: int, allow: torch.Tensor) -> torch.Tensor:
    """Shared body of multi_head_attention_space/_time (model/timesformer_clip.py:55-205)
    written as ONE masked attention over all 1+F*P tokens: the cls query attends to every
    token (:81,:158); a patch query attends to cls + the patches ``allow`` marks (same
    frame :84-108 / same position :161-185).  Softmax over the allowed set is identical
    to the reference's gather-then-softmax."""
    mask = torch.zeros(allow.shape, dtype=x.dtype).masked_fill(~allow, float("-inf"))
    return mha(x, sd, p, heads, mask)


def v1_allow_masks(Fr: int, P: int):
    """Token order is (frames patches): row 1 + t*P + n (model/timesformer_clip.py:84-98)."""
    T = 1 + Fr * P
    t = torch.arange(Fr).repeat_interleave(P)
    n = torch.arange(P).repeat(Fr)
    space = torch.ones(T, T, dtype=torch.bool)
    time = torch.ones(T, T, dtype=torch.bool)
    space[1:, 1:] = t[:, None] == t[None, :]
    time[1:, 1:] = n[:, None] == n[None, :]
    return space, time


def timesformer_v1(video: torch.Tensor, sd: SD, arch: ClipArch, p: str = "visual.") -> torch.Tensor:
    """model/timesformer_clip.py:384-438 + block :308-315."""
    B, Fr = video.shape[:2]
    x = patch_embed(video.reshape(B * Fr, *video.shape[2:]), sd, p)            # [(b t), P, W]
    P, W = x.shape[1], x.shape[2]
    x = x.reshape(B, Fr * P, W)                                                # :392 (frames patches)
    pos = sd[f"{p}positional_embedding"].to(x.dtype)
    tile_pos = pos[1:].repeat(Fr, 1)                                           # :411
    tile_time = sd[f"{p}temporal_embed"].to(x.dtype)[:Fr].repeat_interleave(P, dim=0)  # :415-417
    cls = (sd[f"{p}class_embedding"].to(x.dtype) + pos[0]).expand(B, 1, W)
    x = torch.cat([cls, x + (tile_pos + tile_time)[None]], dim=1)              # :420-424
    x = layer_norm(x, sd[f"{p}ln_pre.weight"], sd[f"{p}ln_pre.bias"])
    space, time = v1_allow_masks(Fr, P)
    for i in range(n_layers(sd, f"{p}transformer")):
        q = f"{p}transformer.resblocks.{i}"
        x = x + _v1_masked_attention(layer_norm(x, sd[f"{q}.ln_time.weight"], sd[f"{q}.ln_time.bias"]),
                                     sd, f"{q}.timeattn", arch.vision_heads, time)   # :309
        x = x + _v1_masked_attention(layer_norm(x, sd[f"{q}.ln_1.weight"], sd[f"{q}.ln_1.bias"]),
                                     sd, f"{q}.attn", arch.vision_heads, space)      # :310
        x = x + mlp(layer_norm(x, sd[f"{q}.ln_2.weight"], sd[f"{q}.ln_2.bias"]), sd, f"{q}.mlp")  # :314
    x = layer_norm(x[:, 0, :], sd[f"{p}ln_post.weight"], sd[f"{p}ln_post.bias"])
    return x @ sd[f"{p}proj"]
