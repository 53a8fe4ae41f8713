"""Parameter containers with the upstream CLIP / TimeSformer state-dict key names.

These modules own parameters only: ``state_dict()`` / ``load_state_dict()`` are the
drop-in contract (strict loading of the reference's checkpoints, evaluation/eval.py:90-91).
They carry no PyTorch forward; the arithmetic is ``vtc_amd.towers`` -> libvtc_hip.so.

Key names follow upstream ``clip/model.py`` [openai/CLIP, un-vendored] and
model/timesformer_clip_alt.py:112-129,223-244 of the reference:
  visual.{conv1.weight,class_embedding,positional_embedding,temporal_embed,ln_pre.*,ln_post.*,proj}
  visual.transformer.resblocks.N.{attn,timeattn}.{in_proj_weight,in_proj_bias,out_proj.*}
  visual.transformer.resblocks.N.{ln_1,ln_2,ln_time}.*, .mlp.{c_fc,c_proj}.*, .temporal_fc.*
  transformer.resblocks.N.*, token_embedding.weight, positional_embedding, ln_final.*,
  text_projection, logit_scale
"""
from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn


@dataclass(frozen=True)
class ClipConfig:
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


# model/timesformer_clip_alt.py:290-310 + upstream model zoo
CONFIGS = {
    "ViT-B/32": ClipConfig(),
    "ViT-B/16": ClipConfig(vision_patch_size=16),
    "ViT-L/14": ClipConfig(embed_dim=768, vision_layers=24, vision_width=1024, vision_patch_size=14,
                           transformer_width=768, transformer_heads=12),
}


class AttentionParams(nn.Module):
    """in_proj_weight [3W,W], in_proj_bias [3W], out_proj Linear (timesformer_clip_alt.py:70-84)."""

    def __init__(self, width: int, std: float, proj_std: float, zero_out=False):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * std)
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)
        nn.init.normal_(self.out_proj.weight, std=proj_std)
        nn.init.zeros_(self.out_proj.bias)


class NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward pass runs in libvtc_hip.so (vtc_amd.towers)")


class BlockParams(NoForward):
    def __init__(self, width: int, layers: int, timesformer, v1: bool = False):
        super().__init__()
        attn_std = width ** -0.5
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        fc_std = (2 * width) ** -0.5
        self.attn = AttentionParams(width, attn_std, proj_std)
        self.ln_1 = nn.LayerNorm(width)
        c_fc, c_proj = nn.Linear(width, 4 * width), nn.Linear(4 * width, width)
        nn.init.normal_(c_fc.weight, std=fc_std)
        nn.init.normal_(c_proj.weight, std=proj_std)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", c_fc), ("gelu", nn.Identity()), ("c_proj", c_proj)]))
        self.ln_2 = nn.LayerNorm(width)
        if timesformer:
            self.timeattn = AttentionParams(width, 0.02, 0.02)
            self.ln_time = nn.LayerNorm(width)
            if v1:   # model/timesformer_clip.py:264-269: time attention starts as a zero map
                nn.init.zeros_(self.timeattn.in_proj_weight)
                nn.init.ones_(self.timeattn.out_proj.weight)
            else:
                self.temporal_fc = nn.Linear(width, width)
                nn.init.zeros_(self.temporal_fc.weight)   # timesformer_clip_alt.py:246-250
                nn.init.zeros_(self.temporal_fc.bias)


class Transformer(NoForward):
    """clip.model.Transformer(width, layers, heads) -- also the CAM (model/model.py:396)."""

    def __init__(self, width: int, layers: int, heads: int, timesformer: bool = False, v1: bool = False):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[BlockParams(width, layers, timesformer, v1) for _ in range(layers)])


class VisionParams(nn.Module):
    """Upstream VisionTransformer (nframes = 0) or the TimeSformer VisualTransformer
    (model/timesformer_clip_alt.py:203-250; v1: model/timesformer_clip.py:341-382).

    ``forward`` is the tower's own entry point -- ``self.model.visual(vis)`` in the reference
    (model/model.py:497,613) and the direct call of model/timesformer_clip_alt.py:333-360: it packs the
    parameters once per (parameter versions, compute dtype) and runs vtc_vision_forward in
    libvtc_hip.so.  Eval only, GPU only, no fallback."""

    #: arithmetic of the GEMM/attention operands (LayerNorm, softmax, residual stream, output stay fp32)
    compute_dtype = torch.bfloat16
    #: multiply temporal_fc and timeattn.out_proj together at pack time (one GEMM instead of two)
    fuse_temporal = True

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, nframes=0, v1=False):
        super().__init__()
        self._packed_sig, self._packed = None, None
        self.input_resolution, self.output_dim, self.nframes, self.width = input_resolution, output_dim, nframes, width
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        if nframes:
            self.temporal_embed = nn.Parameter(torch.zeros(nframes, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads, timesformer=bool(nframes), v1=v1)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[B, F, 3, H, W] (TimeSformer, model/timesformer_clip_alt.py:252-286 / timesformer_clip.py:384-438) or
        [B, 3, H, W] (upstream VisionTransformer.forward) -> [B, output_dim] fp32."""
        if self.training:
            raise RuntimeError("vtc_amd implements the forward/eval path only: call .eval()")
        if not x.is_cuda:
            raise RuntimeError("vtc_amd: inputs must be on the GPU (no CPU fallback)")
        if (x.dim() == 5) != bool(self.nframes):
            raise ValueError(f"expected {'[B,F,3,H,W]' if self.nframes else '[B,3,H,W]'} pixels, got {tuple(x.shape)}")
        from .. import towers
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (self.compute_dtype, self.fuse_temporal)
        if self._packed_sig != sig:
            if next(self.parameters()).device != x.device:
                raise RuntimeError("vtc_amd: module and input are on different devices")
            self._packed = towers.PackedVision(dict(self.state_dict()), "", self.compute_dtype, self.fuse_temporal)
            self._packed_sig = sig
        return self._packed.forward(x)


class VisualTransformer(VisionParams):
    """``model.timesformer_clip_alt.VisualTransformer`` (model/timesformer_clip_alt.py:203-286): same
    constructor signature, parameter names and ``forward([B,F,3,H,W]) -> [B,output_dim]``."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 nframes: int):
        if width != heads * 64:
            raise NotImplementedError(f"head_dim must be 64 on the HIP path (width {width}, heads {heads})")
        super().__init__(input_resolution, patch_size, width, layers, heads, output_dim, nframes=nframes)


class VisualTransformerV1(VisionParams):
    """``model.timesformer_clip.VisualTransformer`` (model/timesformer_clip.py:341-438): the older variant."""

    def __init__(self, input_resolution: int, patch_size: int, width: int, layers: int, heads: int, output_dim: int,
                 nframes: int):
        if width != heads * 64:
            raise NotImplementedError(f"head_dim must be 64 on the HIP path (width {width}, heads {heads})")
        super().__init__(input_resolution, patch_size, width, layers, heads, output_dim, nframes=nframes, v1=True)


class ClipParams(nn.Module):
    """Upstream CLIP container: what ``clip.load`` returns in the reference (model/model.py:317)."""

    def __init__(self, cfg: ClipConfig):
        super().__init__()
        self.cfg = cfg
        self.context_length = cfg.context_length
        self.visual = VisionParams(cfg.image_resolution, cfg.vision_patch_size, cfg.vision_width, cfg.vision_layers,
                                   cfg.vision_width // 64, cfg.embed_dim)
        self.transformer = Transformer(cfg.transformer_width, cfg.transformer_layers, cfg.transformer_heads)
        self.vocab_size = cfg.vocab_size
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.transformer_width)
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        self.positional_embedding = nn.Parameter(torch.empty(cfg.context_length, cfg.transformer_width).normal_(std=0.01))
        self.ln_final = nn.LayerNorm(cfg.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(cfg.transformer_width, cfg.embed_dim).normal_(std=cfg.transformer_width ** -0.5))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    #: arithmetic of the text tower's GEMM/attention operands when called through ``encode_text`` directly
    compute_dtype = torch.bfloat16

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """Upstream ``CLIP.encode_image`` (called at model/model.py:332,335,464,467): the visual tower."""
        return self.visual(image)

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """Upstream ``CLIP.encode_text`` (called at model/model.py:210,340,351,472,499,615) on libvtc_hip.so."""
        if self.training:
            raise RuntimeError("vtc_amd implements the forward/eval path only: call .eval()")
        if not text.is_cuda:
            raise RuntimeError("vtc_amd: inputs must be on the GPU (no CPU fallback)")
        from .. import towers
        own = [p for n, p in self.named_parameters() if not n.startswith("visual.")]
        sig = tuple((p.data_ptr(), p._version) for p in own) + (self.compute_dtype,)
        if getattr(self, "_text_sig", None) != sig:
            self._text_packed = towers.PackedText(dict(self.state_dict()), "", self.compute_dtype, heads=self.transformer.heads)
            self._text_sig = sig
        return self._text_packed.forward(text)


def pretrained_weights_available() -> bool:
    return bool(os.environ.get("VTC_CLIP_WEIGHTS"))


_warned_random = False


def load(model_type: str = "ViT-B/32", device="cpu", cfg: ClipConfig = None) -> ClipParams:
    """Stand-in for ``clip.load(model_type, device="cpu", jit=False)[0]`` (model/model.py:317).

    There is no network here, so the weights are random (upstream-style init) unless the
    environment variable VTC_CLIP_WEIGHTS names a ``torch.save``d upstream state dict."""
    if isinstance(model_type, ClipConfig):
        cfg = model_type
    if cfg is None:
        if model_type not in CONFIGS:
            raise ValueError(f"unknown CLIP model type {model_type!r}; known: {sorted(CONFIGS)}")
        cfg = CONFIGS[model_type]
    m = ClipParams(cfg)
    path = os.environ.get("VTC_CLIP_WEIGHTS")
    if path:
        sd = torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        sd = {k: v.float() for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
        m.load_state_dict(sd, strict=True)
    elif not isinstance(model_type, ClipConfig):
        # a named upstream model without its weights: say so once (a ClipConfig instance is a test/bench architecture)
        global _warned_random
        if not _warned_random:
            import warnings
            warnings.warn(f"vtc_amd: clip_arch.load({model_type!r}) has no pretrained weights (VTC_CLIP_WEIGHTS is unset, and "
                          "there is no network to download them as the reference's clip.load does): the towers are "
                          "RANDOMLY initialised until a checkpoint / state dict is loaded", stacklevel=2)
            _warned_random = True
    return m.to(device).eval()


def make_timesformer_clip_vit_alt(nframes: int, model="ViT-B/32", clip_model: ClipParams = None, cfg: ClipConfig = None):
    """model/timesformer_clip_alt.py:289-330: a TimeSformer tower initialised from the CLIP ViT
    weights; only time/temporal keys may be missing (:325-328)."""
    if cfg is None:
        cfg = clip_model.cfg if clip_model is not None else CONFIGS[model]
    t = VisualTransformer(cfg.image_resolution, cfg.vision_patch_size, cfg.vision_width, cfg.vision_layers,
                          cfg.vision_width // 64, cfg.embed_dim, nframes=nframes)
    if clip_model is None:
        clip_model = load(model, cfg=cfg)
    missing, unexpected = t.load_state_dict(clip_model.visual.state_dict(), strict=False)
    assert len(unexpected) == 0
    assert all(("time" in x or "temporal" in x) for x in missing)
    return t


def make_timesformer_clip_vit(nframes: int, model="ViT-B/32", clip_model: ClipParams = None, cfg: ClipConfig = None):
    """model/timesformer_clip.py:441-467: the older TimeSformer variant (cls attends globally, patches attend
    to cls + same frame / same position; no temporal_fc), initialised from the CLIP ViT weights."""
    if cfg is None:
        cfg = clip_model.cfg if clip_model is not None else CONFIGS[model]
    t = VisualTransformerV1(cfg.image_resolution, cfg.vision_patch_size, cfg.vision_width, cfg.vision_layers,
                            cfg.vision_width // 64, cfg.embed_dim, nframes=nframes)
    if clip_model is None:
        clip_model = load(model, cfg=cfg)
    missing, unexpected = t.load_state_dict(clip_model.visual.state_dict(), strict=False)
    assert len(unexpected) == 0
    assert all(("time" in x or "temporal" in x) for x in missing)
    return t
