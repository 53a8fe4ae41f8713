"""CPU: the oracle's restatement of the un-vendored openai/CLIP towers vs HuggingFace
``transformers`` CLIP -- an independent implementation of the same published architecture
(SURVEY 8c).  Weights are the seeded synthetic ones, mapped by name."""
import numpy as np
import pytest
import torch

from oracle import arch as A
from oracle.clip_ref import encode_image, encode_text

transformers = pytest.importorskip("transformers")
torch.set_grad_enabled(False)


def to_hf(sd, a):
    out = {}

    def blocks(src, dst, n, width):
        for i in range(n):
            s, d = f"{src}.resblocks.{i}", f"{dst}.encoder.layers.{i}"
            w, b = sd[f"{s}.attn.in_proj_weight"], sd[f"{s}.attn.in_proj_bias"]
            for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
                out[f"{d}.self_attn.{nm}.weight"] = w[j * width:(j + 1) * width]
                out[f"{d}.self_attn.{nm}.bias"] = b[j * width:(j + 1) * width]
            out[f"{d}.self_attn.out_proj.weight"] = sd[f"{s}.attn.out_proj.weight"]
            out[f"{d}.self_attn.out_proj.bias"] = sd[f"{s}.attn.out_proj.bias"]
            for ln, hl in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2")):
                out[f"{d}.{hl}.weight"], out[f"{d}.{hl}.bias"] = sd[f"{s}.{ln}.weight"], sd[f"{s}.{ln}.bias"]
            for fc, hf in (("c_fc", "fc1"), ("c_proj", "fc2")):
                out[f"{d}.mlp.{hf}.weight"], out[f"{d}.mlp.{hf}.bias"] = sd[f"{s}.mlp.{fc}.weight"], sd[f"{s}.mlp.{fc}.bias"]

    blocks("visual.transformer", "vision_model", a.vision_layers, a.vision_width)
    blocks("transformer", "text_model", a.transformer_layers, a.transformer_width)
    out["vision_model.embeddings.patch_embedding.weight"] = sd["visual.conv1.weight"]
    out["vision_model.embeddings.class_embedding"] = sd["visual.class_embedding"]
    out["vision_model.embeddings.position_embedding.weight"] = sd["visual.positional_embedding"]
    out["vision_model.pre_layrnorm.weight"], out["vision_model.pre_layrnorm.bias"] = sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"]
    out["vision_model.post_layernorm.weight"], out["vision_model.post_layernorm.bias"] = sd["visual.ln_post.weight"], sd["visual.ln_post.bias"]
    out["visual_projection.weight"] = sd["visual.proj"].t().contiguous()
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    out["text_projection.weight"] = sd["text_projection"].t().contiguous()
    out["logit_scale"] = sd["logit_scale"]
    return out


def test_oracle_matches_hf_clip():
    a = A.TINY
    cfg = transformers.CLIPConfig(
        text_config=dict(vocab_size=a.vocab_size, hidden_size=a.transformer_width, intermediate_size=4 * a.transformer_width,
                         num_hidden_layers=a.transformer_layers, num_attention_heads=a.transformer_heads,
                         max_position_embeddings=a.context_length, hidden_act="quick_gelu", eos_token_id=A.EOT,
                         bos_token_id=A.SOT, pad_token_id=0, projection_dim=a.embed_dim),
        vision_config=dict(hidden_size=a.vision_width, intermediate_size=4 * a.vision_width,
                           num_hidden_layers=a.vision_layers, num_attention_heads=a.vision_heads,
                           image_size=a.image_resolution, patch_size=a.vision_patch_size, hidden_act="quick_gelu",
                           projection_dim=a.embed_dim),
        projection_dim=a.embed_dim)
    hf = transformers.CLIPModel(cfg).eval()
    sd = {}
    sd.update(A.synth_visual(a, 41, prefix="visual."))
    sd.update(A.synth_text(a, 42))
    missing, unexpected = hf.load_state_dict(to_hf(sd, a), strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    img = A.synth_pixels((3, 3, a.image_resolution, a.image_resolution), 43)
    txt = A.synth_tokens(5, a, 44, empty_frac=0.2)
    def feats(x):
        return x if isinstance(x, torch.Tensor) else x.pooler_output
    hv = feats(hf.get_image_features(pixel_values=img))
    ht = feats(hf.get_text_features(input_ids=txt, attention_mask=torch.ones_like(txt)))
    np.testing.assert_allclose(encode_image(img, sd, a).numpy(), hv.numpy(), atol=2e-5)
    np.testing.assert_allclose(encode_text(txt, sd, a).numpy(), ht.numpy(), atol=2e-5)
