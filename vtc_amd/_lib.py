"""ctypes binding of libvtc_hip.so (include/vtc_hip.h).

The library is the product; there is NO fallback.  If it is missing or a symbol
does not resolve, importing/using the ops raises -- loudly -- instead of silently
running something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VTC_HIP_LIB: kernel-tuning experiments only (a differently compiled build of the same sources)
LIB_PATH = os.environ.get("VTC_HIP_LIB") or os.path.join(_HERE, "lib", "libvtc_hip.so")

VTC_F32, VTC_BF16, VTC_U8 = 0, 1, 2
ACT_NONE, ACT_NORMALIZE, ACT_SQUASH, ACT_TANH, ACT_SUB_MEAN, ACT_BN = 0, 1, 2, 3, 4, 5
SWEEP_F32, SWEEP_BF16X3, SWEEP_BF16, SWEEP_EXACT = 0, 1, 2, 3
EPI_STORE, EPI_GELU, EPI_RESID = 0, 1, 2
PROF_CLASSES = ("gemm_bf16", "gemm_f32", "attention", "norm", "embed", "topk")
PROF_REGIONS = ("other", "attn", "mlp")
VTC_F16 = 3
# vtc_vision_w.flags / vtc_text_w.flags (include/vtc_hip.h VTC_TOWER_*): per-model path switches
TOWER_NO_LN_FOLD, TOWER_FULL_LAST_LAYER, TOWER_NO_SPLITK = 1, 8, 16
CAM_NO_FUSED = 1
RECALL_NONFINITE = 1 << 40      # include/vtc_hip.h VTC_RECALL_NONFINITE: ORed into the first counter of a direction by the recall-only sweeps
ABI_VERSION = 7

vp, fp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers


class BlockW(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b", "ln2_g", "ln2_b", "fc_w", "fc_b", "proj_w", "proj_b",
        "lnt_g", "lnt_b", "tqkv_w", "tqkv_b", "tout_w", "tout_b", "tfc_w", "tfc_b",
        "qkv_wf", "qkv_s", "qkv_c", "fc_wf", "fc_s", "fc_c", "tqkv_wf", "tqkv_s", "tqkv_c")]


class VisionW(C.Structure):
    _fields_ = [("width", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("patch", C.c_int), ("grid", C.c_int),
                ("embed_dim", C.c_int), ("nframes", C.c_int), ("variant", C.c_int), ("flags", C.c_int),
                ("pix_mean", C.c_float * 3), ("pix_std", C.c_float * 3),
                ("conv_w", C.c_void_p), ("class_embedding", C.c_void_p), ("pos", C.c_void_p), ("temporal", C.c_void_p),
                ("ln_pre_g", C.c_void_p), ("ln_pre_b", C.c_void_p), ("ln_post_g", C.c_void_p), ("ln_post_b", C.c_void_p),
                ("proj_t", C.c_void_p), ("blocks", C.POINTER(BlockW))]


class TextW(C.Structure):
    _fields_ = [("width", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("ctx", C.c_int), ("vocab", C.c_int),
                ("embed_dim", C.c_int), ("half_layers", C.c_int), ("flags", C.c_int),
                ("tok_emb", C.c_void_p), ("pos", C.c_void_p), ("ln_final_g", C.c_void_p), ("ln_final_b", C.c_void_p),
                ("proj_t", C.c_void_p), ("blocks", C.POINTER(BlockW))]


class CamW(C.Structure):
    _fields_ = [("width", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("init_from_avg", C.c_int),
                ("residual_activation", C.c_int), ("squash_scale", C.c_float),
                ("final_linear", C.c_void_p), ("mask_embedding", C.c_void_p), ("blocks", C.POINTER(BlockW)),
                ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p), ("flags", C.c_int)]


# name -> (restype, argtypes); must list EVERY symbol include/vtc_hip.h declares
SIGNATURES = {
    "vtc_last_error": (C.c_char_p, []),
    "vtc_abi_version": (C.c_int, []),
    "vtc_vision_workspace_bytes": (C.c_size_t, [C.POINTER(VisionW), C.c_int, C.c_int, C.c_int]),
    "vtc_vision_forward": (C.c_int, [C.POINTER(VisionW), vp, C.c_int, C.c_int, C.c_int, fp, vp, C.c_size_t, C.c_int, vp]),
    "vtc_text_workspace_bytes": (C.c_size_t, [C.POINTER(TextW), C.c_int, C.c_int]),
    "vtc_text_forward": (C.c_int, [C.POINTER(TextW), ip, C.c_int, fp, vp, C.c_size_t, C.c_int, vp]),
    "vtc_text_forward2": (C.c_int, [C.POINTER(TextW), ip, C.c_int, ip, C.c_int, C.c_int, fp, vp, C.c_size_t, C.c_int, vp]),
    "vtc_text_ragged_workspace_bytes": (C.c_size_t, [C.POINTER(TextW), C.c_int, C.c_int, C.c_int]),
    "vtc_text_forward_ragged": (C.c_int, [C.POINTER(TextW), ip, C.c_int, ip, C.c_int, fp, vp, C.c_size_t, C.c_int, vp]),
    "vtc_cam_workspace_bytes": (C.c_size_t, [C.POINTER(CamW), C.c_int, C.c_int, C.c_int]),
    "vtc_cam_forward": (C.c_int, [C.POINTER(CamW), fp, fp, ip, C.c_int, C.c_int, C.c_int, fp, vp, C.c_size_t, C.c_int, vp]),
    "vtc_cam_fused_gave_up": (C.c_int, [C.c_int]),
    "vtc_normalize_rows": (C.c_int, [fp, fp, C.c_int, C.c_int, vp]),
    "vtc_normalize_rows2": (C.c_int, [fp, fp, C.c_int, fp, fp, C.c_int, C.c_int, vp, vp]),
    "vtc_mean_groups": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "vtc_nonfinite_flag": (C.c_int, [fp, C.c_size_t, vp, vp]),
    "vtc_nonfinite_flag2": (C.c_int, [fp, C.c_size_t, fp, C.c_size_t, vp, vp]),
    "vtc_mean_head_groups": (C.c_int, [fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "vtc_pack_tokens": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vtc_single_query_attention": (C.c_int, [vp, vp, fp] + [C.c_int] * 9 + [vp, vp, C.c_int, C.c_int, vp]),
    "vtc_segment_mean": (C.c_int, [fp, ip, fp, C.c_int, C.c_int, vp]),
    "vtc_similarity": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, fp, fp, vp]),
    "vtc_clip_loss_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vtc_clip_loss": (C.c_int, [fp, C.c_int, fp, vp, C.c_size_t, vp]),
    "vtc_l2_topk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtc_l2_topk": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp, vp, C.c_size_t, vp]),
    "vtc_l2_topk_bidir_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtc_l2_topk_bidir": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp, ip, fp, vp, C.c_size_t, vp]),
    "vtc_l2_sweep_row_block": (C.c_int, []),
    "vtc_l2_sweep_shard_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "vtc_l2_sweep_shard_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "vtc_l2_sweep_shard_rows": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, ip, fp, vp, C.c_int, vp, C.c_size_t, vp]),
    "vtc_l2_sweep_shard_cols": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, ip, fp, vp,
                                          C.c_size_t, vp]),
    "vtc_recall_hits": (C.c_int, [ip, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int), C.c_int, vp, vp]),
    "vtc_recall_hits_pair": (C.c_int, [ip, ip, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int), C.c_int, vp, vp, vp]),
    "vtc_l2_recall_bidir_supported": (C.c_int, [C.c_int, C.c_int]),
    "vtc_l2_recall_bidir_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "vtc_l2_recall_bidir": (C.c_int, [fp, fp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, ip, ip, vp, C.c_size_t, vp]),
    "vtc_l2_recall_shard_supported": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "vtc_l2_recall_planes": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int]),
    "vtc_l2_recall_shard_rows": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, ip, vp, C.c_int, vp, C.c_size_t, vp]),
    "vtc_l2_recall_shard_cols": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, vp, C.c_int, C.c_int, vp, ip, vp,
                                           C.c_size_t, vp]),
    "vtc_gemm": (C.c_int, [vp, vp, fp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "vtc_gemm_resid_layernorm_workspace_bytes": (C.c_size_t, [C.c_int]),
    "vtc_gemm_resid_layernorm_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vtc_gemm_resid_layernorm": (C.c_int, [vp, vp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, vp, vp, C.c_size_t, vp]),
    "vtc_layernorm": (C.c_int, [fp, fp, fp, vp, C.c_int, C.c_int, C.c_int, ip, C.c_int, vp]),
    "vtc_prof_begin": (C.c_int, []),
    "vtc_prof_end": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "vtc_prof_end_regions": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "vtc_prof_end_records": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "vtc_debug_launch_count": (C.c_longlong, []),
    "vtc_attention": (C.c_int, [vp, vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, vp]),
    # adapter-only training step (backward + optimizer primitives)
    "vtc_transpose_f32": (C.c_int, [fp, fp, C.c_int, C.c_int, vp]),
    "vtc_colsum_f32": (C.c_int, [fp, fp, C.c_int, C.c_int, vp]),
    "vtc_layernorm_bwd": (C.c_int, [fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "vtc_attention_small_bwd": (C.c_int, [fp, fp, fp, C.c_int, C.c_int, C.c_int, vp]),
    "vtc_quickgelu": (C.c_int, [fp, fp, fp, C.c_size_t, vp]),
    "vtc_normalize_rows_bwd": (C.c_int, [fp, fp, fp, C.c_int, C.c_int, vp]),
    "vtc_clip_loss_bwd": (C.c_int, [fp, C.c_int, fp, vp, C.c_size_t, vp]),
    "vtc_adam_step": (C.c_int, [fp, fp, fp, fp, fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp]),
    "vtc_axpby": (C.c_int, [fp, fp, fp, C.c_float, C.c_float, C.c_size_t, vp]),
    "vtc_scale_rows": (C.c_int, [fp, fp, C.c_int, C.c_int, C.c_int, vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libvtc_hip.so once; raise if it (or any declared symbol) is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libvtc_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C vtc_amd/csrc`).  vtc_amd has no CPU or PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)  # AttributeError if the symbol is not exported
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().vtc_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed: {msg}")
