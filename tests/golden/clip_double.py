"""Tests-only stand-in for the un-vendored ``clip`` package (openai/CLIP).

Used ONLY by tests/golden/make_golden.py, in this container, so that the
reference's own ``model/model.py`` / ``model/timesformer_clip*.py`` import and
run UNMODIFIED and emit golden vectors.  It is never shipped,
never imported by the product, and never runs on the GPU box.

It provides exactly what the reference touches:
  clip.load(name, device=, jit=) -> (model, preprocess)      model/model.py:317,392,486,555
  clip.model.Transformer(width=, layers=, heads=)            model/model.py:396,560
  model.encode_image / encode_text / visual / transformer / ln_final /
  logit_scale / dtype / float()                              model/model.py:318-369
built on ``torch.nn.MultiheadAttention`` (sequence-first), i.e. on torch's own
attention rather than on oracle/clip_ref.py's hand-written one, which makes the
golden vectors an independent check of the oracle's upstream restatement too.
Random init; make_golden.py then loads the seeded synthetic state dict strictly.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

ARCH = None  # set by make_golden.py to an oracle.arch.ClipArch before clip.load()


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        t = x.dtype
        return super().forward(x.type(torch.float32)).type(t)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask

    def attention(self, x):
        m = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        return self.attn(x, x, x, need_weights=False, attn_mask=m)[0]

    def forward(self, x):
        x = x + self.attention(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


class VisionTransformer(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))

    def forward(self, x):
        x = self.conv1(x)
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
        cls = self.class_embedding.to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1) + self.positional_embedding.to(x.dtype)
        x = self.ln_pre(x).permute(1, 0, 2)
        x = self.transformer(x).permute(1, 0, 2)
        x = self.ln_post(x[:, 0, :])
        return x @ self.proj


class CLIP(nn.Module):
    def __init__(self, a):
        super().__init__()
        self.context_length = a.context_length
        self.visual = VisionTransformer(a.image_resolution, a.vision_patch_size, a.vision_width, a.vision_layers,
                                        a.vision_width // 64, a.embed_dim)
        mask = torch.empty(a.context_length, a.context_length).fill_(float("-inf")).triu_(1)
        self.transformer = Transformer(a.transformer_width, a.transformer_layers, a.transformer_heads, mask)
        self.vocab_size = a.vocab_size
        self.token_embedding = nn.Embedding(a.vocab_size, a.transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(a.context_length, a.transformer_width).normal_(std=0.01))
        self.ln_final = LayerNorm(a.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(a.transformer_width, a.embed_dim).normal_(std=0.05))
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        return self.visual(image.type(self.dtype))

    def encode_text(self, text):
        x = self.token_embedding(text).type(self.dtype)
        x = x + self.positional_embedding.type(self.dtype)
        x = self.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)
        x = self.ln_final(x).type(self.dtype)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


def load(name, device="cpu", jit=False):
    assert ARCH is not None, "set clip_double.ARCH first"
    return CLIP(ARCH).eval(), None


def install():
    """Register this module as ``clip`` (with a ``clip.model`` submodule) and an empty ``faiss``."""
    me = sys.modules[__name__]
    sys.modules["clip"] = me
    sub = types.ModuleType("clip.model")
    sub.Transformer = Transformer
    sys.modules["clip.model"] = sub
    me.model = sub
    sys.modules.setdefault("faiss", types.ModuleType("faiss"))
