"""CPU: the C-ABI library loads and exports every symbol include/vtc_hip.h declares; the host
mirror keeps the reference's plugin contract (class names, ctor kwargs, state-dict keys)."""
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vtc_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "vtc_hip.h")).read()
    declared = set(re.findall(r"\b(vtc_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vtc_block_w", "vtc_vision_w", "vtc_text_w", "vtc_cam_w"}
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.lib()                       # raises if the .so or any symbol is missing
    assert lib.vtc_abi_version() == 2
    for name in declared:
        assert hasattr(lib, name)


def test_argument_errors_are_reported_not_thrown():
    from vtc_amd import _lib as L
    lib = L.lib()
    rc = lib.vtc_gemm(None, None, None, None, 0, 0, 0, 0, 0, 0, 0, None)    # empty problem: rejected before any launch
    assert rc != 0 and b"gemm" in lib.vtc_last_error()
    rc = lib.vtc_recall_hits(None, 1, 11, 0, (L.C.c_int * 1)(12), 1, None, None)
    assert rc != 0 and b"recall_hits" in lib.vtc_last_error()


def test_state_dict_contract_matches_reference_keys():
    """The synthetic state dicts were loaded STRICTLY into the reference's own classes when the
    golden vectors were made; loading the same dicts strictly here pins the key/shape contract."""
    from dataclasses import asdict

    from oracle import arch as A
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    cfg = ClipConfig(**asdict(A.TINY))
    for kind, cls in (("clip", HM.PretrainedCLIP), ("clip_finaltf", HM.PretrainedCLIP_finaltf),
                      ("timesformer", HM.PretrainedCLIP_TimeSformer), ("timesformer_finaltf", HM.PretrainedCLIP_TimeSformer_finaltf)):
        m = cls(model_type=cfg)
        m.load_state_dict(A.synth_model(A.TINY, 1, kind), strict=True)
    # constructor kwargs of the reference (model/model.py:309-315,375-390,484,540-553)
    want = {"PretrainedCLIP": ["model_type", "freeze", "residual_activation", "comment_fusion"],
            "PretrainedCLIP_finaltf": ["model_type", "freeze", "branch_to_adapt", "branch_to_adapt_val", "residual_activation",
                                       "n_layers", "n_heads", "init_from_avg", "random_comment_masking", "random_skip_adapter",
                                       "init_audio_model", "audio_model_ckpt", "clip_audio_ckpt"],
            "PretrainedCLIP_TimeSformer": ["model_type", "freeze", "residual_activation"],
            "PretrainedCLIP_TimeSformer_finaltf": ["model_type", "freeze", "branch_to_adapt", "branch_to_adapt_val",
                                                   "residual_activation", "visual_device", "n_layers", "n_heads", "init_from_avg",
                                                   "random_comment_masking", "random_skip_adapter"]}
    for name, args in want.items():
        got = list(inspect.signature(getattr(HM, name).__init__).parameters)[1:]
        assert got == args, (name, got)


def test_cam_init_from_avg_zeroing():
    from dataclasses import asdict

    from oracle import arch as A
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    m = HM.PretrainedCLIP_finaltf(model_type=ClipConfig(**asdict(A.TINY)))
    for blk in m.final_transformer.resblocks:                      # model/model.py:440-450
        assert blk.mlp.c_proj.weight.abs().sum() == 0 and blk.attn.out_proj.weight.abs().sum() == 0
    assert m.final_linear.weight.abs().sum() == 0                  # :452


def test_reference_configs_load(tmp_path):
    from vtc_amd.host.parse_config import ConfigParser, read_jsonc
    ref = "/root/reference/configs"
    paths = [os.path.join(ROOT, "configs", f) for f in os.listdir(os.path.join(ROOT, "configs"))]
    if os.path.isdir(ref):
        paths += [os.path.join(ref, f) for f in os.listdir(ref) if f.endswith(".jsonc")]
    assert paths
    for p in paths:
        cfg = read_jsonc(p)
        assert "arch" in cfg and "type" in cfg["arch"]
    c = ConfigParser.from_file(paths[0], modification={"arch;args;branch_to_adapt_val": "skip", "batch_size": 7})
    assert c["batch_size"] == 7
