"""Architecture constants + seeded synthetic weights (TEST INFRASTRUCTURE ONLY).

No checkpoint can be downloaded here, so every parity case runs on weights that
are *generated from a seed* by a pure-numpy generator.  The generator is shared
by the golden-vector script (which feeds the reference's own Python) and by the
GPU parity tests (which feed the HIP path), so fixtures hold only inputs and
expected outputs, never weights.

Key names are the drop-in state-dict contract:
  * upstream CLIP names [openai/CLIP clip/model.py, un-vendored]:
    ``visual.conv1.weight``, ``visual.class_embedding``, ``transformer.resblocks.N.*`` ...
  * TimeSformer additions: model/timesformer_clip_alt.py:112-129, :232-244
    (``timeattn``, ``ln_time``, ``temporal_fc``, ``temporal_embed``)
  * CAM: model/model.py:396-400 (``final_transformer``, ``final_linear``, ``mask_embedding``)
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict

import numpy as np
import torch


@dataclass(frozen=True)
class ClipArch:
    embed_dim: int = 512
    image_resolution: int = 224
    vision_layers: int = 12
    vision_width: int = 768
    vision_patch_size: int = 32
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12

    @property
    def vision_heads(self) -> int:  # upstream: vision_width // 64
        return self.vision_width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def patches(self) -> int:
        return self.grid * self.grid


# ViT-B/32: model/timesformer_clip_alt.py:290-296 + upstream CLIP defaults.
VIT_B32 = ClipArch()
# ViT-B/16 and ViT-L/14: model/timesformer_clip_alt.py:297-310 (197 / 257 tokens per frame; upstream text towers 512 x 8 / 768 x 12).
VIT_B16 = ClipArch(vision_patch_size=16)
VIT_L14 = ClipArch(embed_dim=768, vision_layers=24, vision_width=1024, vision_patch_size=14,
                   transformer_width=768, transformer_heads=12)
# A small architecture with the same structure, for second-scale oracle cases.
# head_dim stays 64 (upstream: heads = width // 64); EOT stays 49407 because
# model/model.py:208 hard-codes it, so the vocabulary keeps its real size.
TINY = ClipArch(embed_dim=128, image_resolution=64, vision_layers=2, vision_width=128,
                vision_patch_size=32, context_length=24, vocab_size=49408,
                transformer_width=128, transformer_heads=2, transformer_layers=2)

SOT, EOT = 49406, 49407


class _Gen:
    """Deterministic N(0, std) tensors from one PCG64 stream (order matters)."""

    def __init__(self, seed: int):
        self.rng = np.random.Generator(np.random.PCG64(seed))

    def normal(self, shape, std):
        a = self.rng.standard_normal(size=shape, dtype=np.float32)
        return torch.from_numpy(a * np.float32(std))

    def ln(self, width):
        # gamma around 1, beta around 0: LayerNorm is exercised with non-trivial affine
        return (torch.from_numpy(1.0 + 0.1 * self.rng.standard_normal(width, dtype=np.float32)),
                torch.from_numpy(0.1 * self.rng.standard_normal(width, dtype=np.float32)))


def _block(g: _Gen, sd: Dict[str, torch.Tensor], p: str, width: int, layers: int,
           timesformer: bool, zero_cam_init: bool = False):
    # upstream CLIP.initialize_parameters std's: attn_std = W^-0.5,
    # proj_std = W^-0.5 (2L)^-0.5, fc_std = (2W)^-0.5.  Biases get a small
    # non-zero std so that bias handling is actually tested.
    attn_std = width ** -0.5
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    names = ["attn"] + (["timeattn"] if timesformer else [])
    for a in names:
        sd[f"{p}.{a}.in_proj_weight"] = g.normal((3 * width, width), attn_std)
        sd[f"{p}.{a}.in_proj_bias"] = g.normal((3 * width,), 0.02)
        sd[f"{p}.{a}.out_proj.weight"] = g.normal((width, width), proj_std)
        sd[f"{p}.{a}.out_proj.bias"] = g.normal((width,), 0.02)
    lns = ["ln_1", "ln_2"] + (["ln_time"] if timesformer else [])
    for n in lns:
        sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"] = g.ln(width)
    sd[f"{p}.mlp.c_fc.weight"] = g.normal((4 * width, width), fc_std)
    sd[f"{p}.mlp.c_fc.bias"] = g.normal((4 * width,), 0.02)
    sd[f"{p}.mlp.c_proj.weight"] = g.normal((width, 4 * width), proj_std)
    sd[f"{p}.mlp.c_proj.bias"] = g.normal((width,), 0.02)
    if timesformer:
        # reference zero-inits temporal_fc (timesformer_clip_alt.py:246-250); a trained
        # checkpoint does not, and a zero temporal branch would test nothing.
        sd[f"{p}.temporal_fc.weight"] = g.normal((width, width), proj_std)
        sd[f"{p}.temporal_fc.bias"] = g.normal((width,), 0.02)
    if zero_cam_init:
        # model/model.py:440-450 (init_from_avg): c_proj.{weight,bias}, attn.out_proj.weight = 0
        sd[f"{p}.mlp.c_proj.weight"].zero_()
        sd[f"{p}.mlp.c_proj.bias"].zero_()
        sd[f"{p}.attn.out_proj.weight"].zero_()
        # nn.MultiheadAttention initialises out_proj.bias to 0, so at init the block is an identity
        sd[f"{p}.attn.out_proj.bias"].zero_()


def synth_visual(arch: ClipArch, seed: int, nframes: int = 0, prefix: str = "",
                 variant: str = "alt") -> Dict[str, torch.Tensor]:
    """Vision tower weights.  nframes == 0 -> upstream ViT; > 0 -> TimeSformer keys too.
    variant "v1" = model/timesformer_clip.py, which has no ``temporal_fc`` (:283-315)."""
    g = _Gen(seed)
    W = arch.vision_width
    scale = W ** -0.5
    sd: Dict[str, torch.Tensor] = {}
    sd["conv1.weight"] = g.normal((W, 3, arch.vision_patch_size, arch.vision_patch_size),
                                  (3 * arch.vision_patch_size ** 2) ** -0.5)
    sd["class_embedding"] = g.normal((W,), scale)
    sd["positional_embedding"] = g.normal((arch.patches + 1, W), scale)
    if nframes:
        sd["temporal_embed"] = g.normal((nframes, W), scale)
    sd["ln_pre.weight"], sd["ln_pre.bias"] = g.ln(W)
    for i in range(arch.vision_layers):
        _block(g, sd, f"transformer.resblocks.{i}", W, arch.vision_layers, timesformer=bool(nframes))
    sd["ln_post.weight"], sd["ln_post.bias"] = g.ln(W)
    sd["proj"] = g.normal((W, arch.embed_dim), scale)
    if variant == "v1":
        sd = {k: v for k, v in sd.items() if "temporal_fc" not in k}
    return {prefix + k: v for k, v in sd.items()}


def synth_text(arch: ClipArch, seed: int, prefix: str = "") -> Dict[str, torch.Tensor]:
    g = _Gen(seed)
    W = arch.transformer_width
    sd: Dict[str, torch.Tensor] = {}
    sd["token_embedding.weight"] = g.normal((arch.vocab_size, W), 0.02)
    sd["positional_embedding"] = g.normal((arch.context_length, W), 0.01)
    for i in range(arch.transformer_layers):
        _block(g, sd, f"transformer.resblocks.{i}", W, arch.transformer_layers, timesformer=False)
    sd["ln_final.weight"], sd["ln_final.bias"] = g.ln(W)
    sd["text_projection"] = g.normal((W, arch.embed_dim), W ** -0.5)
    sd["logit_scale"] = torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float32)
    return {prefix + k: v for k, v in sd.items()}


def synth_cam(arch: ClipArch, seed: int, n_layers: int = 2, init_from_avg_zero: bool = False,
              prefix: str = "") -> Dict[str, torch.Tensor]:
    """CAM weights (model/model.py:396-400).  ``init_from_avg_zero`` reproduces the
    constructor's zeroing (:440-452) -- the at-init state the reference's test pins."""
    g = _Gen(seed)
    D = arch.embed_dim
    sd: Dict[str, torch.Tensor] = {}
    for i in range(n_layers):
        _block(g, sd, f"final_transformer.resblocks.{i}", D, n_layers, timesformer=False,
               zero_cam_init=init_from_avg_zero)
    sd["final_linear.weight"] = torch.zeros(D, D) if init_from_avg_zero else g.normal((D, D), D ** -0.5)
    sd["mask_embedding"] = g.normal((1, D), 1.0)
    return {prefix + k: v for k, v in sd.items()}


def synth_bn_stats(arch: ClipArch, seed: int) -> Dict[str, torch.Tensor]:
    """Buffers of ``mean_center_bn = nn.BatchNorm1d(D, affine=False, momentum=0.2)`` (model/model.py:134-139), as a
    checkpoint trained with residual_activation in {"sub_mean", "bn"} carries them."""
    g = _Gen(seed)
    D = arch.embed_dim
    return {"mean_center_bn.running_mean": g.normal((D,), 0.05), "mean_center_bn.running_var": g.normal((D,), 0.02).abs() + 0.01,
            "mean_center_bn.num_batches_tracked": torch.tensor(7, dtype=torch.long)}


def synth_model(arch: ClipArch, seed: int, kind: str, nframes: int = 8, cam_layers: int = 2,
                cam_at_init: bool = False, bn_stats: bool = False) -> Dict[str, torch.Tensor]:
    """Full state dict for one of the four wrappers (model/model.py:308,374,483,539).

    kind in {"clip", "clip_finaltf", "timesformer", "timesformer_finaltf"}."""
    sd: Dict[str, torch.Tensor] = {}
    tf = kind.startswith("timesformer")
    sd.update(synth_visual(arch, seed * 7 + 1, nframes=nframes if tf else 0, prefix="model.visual."))
    sd.update(synth_text(arch, seed * 7 + 2, prefix="model."))
    if kind.endswith("finaltf"):
        sd.update(synth_cam(arch, seed * 7 + 3, n_layers=cam_layers, init_from_avg_zero=cam_at_init))
    if bn_stats:
        sd.update(synth_bn_stats(arch, seed * 7 + 4))
    return sd


def synth_tokens(n: int, arch: ClipArch, seed: int, empty_frac: float = 0.0) -> torch.Tensor:
    """[n, context] int64 rows ``[SOT, t_1..t_L, EOT, 0...]`` (SURVEY 8d synthetic inputs);
    a fraction of rows is the empty string ``[SOT, EOT, 0...]`` (model/model.py:208)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ctx = arch.context_length
    out = np.zeros((n, ctx), dtype=np.int64)
    lens = rng.integers(1, ctx - 1, size=n)  # 1 .. ctx-2
    empty = rng.random(n) < empty_frac
    for i in range(n):
        L = 0 if empty[i] else int(lens[i])
        out[i, 0] = SOT
        out[i, 1:1 + L] = rng.integers(1, SOT, size=L)
        out[i, 1 + L] = EOT
    return torch.from_numpy(out)


def synth_pixels(shape, seed: int) -> torch.Tensor:
    """N(0,1) fp32 pixels (the distribution tests/test_pretrained_clip.py:8 uses)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal(size=tuple(shape), dtype=np.float32))


def with_dtype(sd: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def sub(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


__all__ = ["ClipArch", "VIT_B32", "TINY", "SOT", "EOT", "synth_visual", "synth_text", "synth_cam",
           "synth_model", "synth_tokens", "with_dtype", "sub", "replace"]
