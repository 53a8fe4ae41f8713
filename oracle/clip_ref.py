"""CPU restatement of the upstream CLIP towers the reference calls into.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The arithmetic lives in the third-party package ``clip`` =
``git+https://github.com/openai/CLIP.git`` (environment.yml:30, un-pinned HEAD),
which is NOT under the reference tree.  This file restates its published
architecture (clip/model.py: LayerNorm, QuickGELU, ResidualAttentionBlock,
Transformer, VisionTransformer, CLIP.encode_text) and is anchored on the
reference's call sites:
  encode_image  model/model.py:332,335,464,467
  encode_text   model/model.py:210,340,351,472,499,615
  clip.model.Transformer(width, layers, heads)  model/model.py:396,560  (the CAM)
The LayerNorm / QuickGELU / MHA math is the same one the reference re-implements
first-party in model/timesformer_clip_alt.py:22-67, which the golden vectors pin.
All functions are functional over a state dict ``sd`` (name -> tensor).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .arch import ClipArch

SD = Dict[str, torch.Tensor]


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """fp32 compute, cast back (model/timesformer_clip_alt.py:22-28; upstream LayerNorm)."""
    if x.dtype == torch.float64:
        return F.layer_norm(x, (x.shape[-1],), w, b, eps)
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x) (model/timesformer_clip_alt.py:31-33)."""
    return x * torch.sigmoid(1.702 * x)


def mha(x: torch.Tensor, sd: SD, p: str, heads: int, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batch-first multi-head self-attention, x [b, L, W].

    Same math as ``nn.MultiheadAttention`` (upstream) and as the reference's own
    ``multi_head_attention`` (model/timesformer_clip_alt.py:43-67): packed q,k,v
    projection, q scaled by head_dim**-0.5 (:52), per-head softmax(q k^T [+mask]) v
    (:36-40), heads merged, out-projection (:65)."""
    b, L, W = x.shape
    hd = W // heads
    qkv = x @ sd[f"{p}.in_proj_weight"].t() + sd[f"{p}.in_proj_bias"]
    q, k, v = qkv.chunk(3, dim=-1)
    q = q * (float(hd) ** -0.5)
    q = q.reshape(b, L, heads, hd).transpose(1, 2)
    k = k.reshape(b, L, heads, hd).transpose(1, 2)
    v = v.reshape(b, L, heads, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    a = s.softmax(dim=-1) @ v
    a = a.transpose(1, 2).reshape(b, L, W)
    return a @ sd[f"{p}.out_proj.weight"].t() + sd[f"{p}.out_proj.bias"]


def mlp(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """c_proj(QuickGELU(c_fc(x))) (model/timesformer_clip_alt.py:115-123)."""
    h = quick_gelu(x @ sd[f"{p}.c_fc.weight"].t() + sd[f"{p}.c_fc.bias"])
    return h @ sd[f"{p}.c_proj.weight"].t() + sd[f"{p}.c_proj.bias"]


def resblock(x: torch.Tensor, sd: SD, p: str, heads: int, mask=None) -> torch.Tensor:
    """upstream ResidualAttentionBlock: x += MHA(ln_1 x); x += MLP(ln_2 x)."""
    x = x + mha(layer_norm(x, sd[f"{p}.ln_1.weight"], sd[f"{p}.ln_1.bias"]), sd, f"{p}.attn", heads, mask)
    x = x + mlp(layer_norm(x, sd[f"{p}.ln_2.weight"], sd[f"{p}.ln_2.bias"]), sd, f"{p}.mlp")
    return x


def n_layers(sd: SD, p: str) -> int:
    pre = f"{p}.resblocks."
    return 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))


def transformer(x: torch.Tensor, sd: SD, p: str, heads: int, mask=None) -> torch.Tensor:
    """upstream Transformer (nn.Sequential of blocks).  x is batch-first here; upstream
    permutes to sequence-first around the call, which does not change the math."""
    for i in range(n_layers(sd, p)):
        x = resblock(x, sd, f"{p}.resblocks.{i}", heads, mask)
    return x


def patch_embed(img: torch.Tensor, sd: SD, p: str = "") -> torch.Tensor:
    """conv1 (k = s = patch, no bias) -> [N, grid*grid, W] (timesformer_clip_alt.py:255-260)."""
    w = sd[f"{p}conv1.weight"]
    x = F.conv2d(img, w, stride=w.shape[-1])
    return x.flatten(2).transpose(2, 1)


def encode_image(img: torch.Tensor, sd: SD, arch: ClipArch, p: str = "visual.") -> torch.Tensor:
    """upstream VisionTransformer.forward: conv1 -> +cls -> +pos -> ln_pre -> blocks
    -> ln_post(x[:,0]) @ proj.  img [N,3,H,W] -> [N, embed_dim]."""
    x = patch_embed(img, sd, p)
    cls = sd[f"{p}class_embedding"].to(x.dtype).expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[f"{p}positional_embedding"].to(x.dtype)
    x = layer_norm(x, sd[f"{p}ln_pre.weight"], sd[f"{p}ln_pre.bias"])
    x = transformer(x, sd, f"{p}transformer", arch.vision_heads)
    x = layer_norm(x[:, 0, :], sd[f"{p}ln_post.weight"], sd[f"{p}ln_post.bias"])
    return x @ sd[f"{p}proj"]


def causal_mask(L: int, dtype) -> torch.Tensor:
    """upstream build_attention_mask: full(-inf).triu_(1)."""
    return torch.full((L, L), float("-inf"), dtype=dtype).triu_(1)


def encode_text(text: torch.Tensor, sd: SD, arch: ClipArch, p: str = "") -> torch.Tensor:
    """upstream CLIP.encode_text: token_embedding + positional_embedding -> causal
    blocks -> ln_final -> row at text.argmax(-1) (EOT is the largest id) @ text_projection.
    text [S, ctx] int64 -> [S, embed_dim]."""
    x = sd[f"{p}token_embedding.weight"][text] + sd[f"{p}positional_embedding"]
    x = transformer(x, sd, f"{p}transformer", arch.transformer_heads, causal_mask(text.shape[1], x.dtype))
    x = layer_norm(x, sd[f"{p}ln_final.weight"], sd[f"{p}ln_final.bias"])
    return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ sd[f"{p}text_projection"]
