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
    assert lib.vtc_abi_version() == L.ABI_VERSION == 7
    for name in declared:
        assert hasattr(lib, name)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The weight structs travel by pointer: the ctypes mirrors (vtc_amd/_lib.py) must have the C compiler's layout of
    include/vtc_hip.h -- sizes and the offsets of the fields added in ABI 5 (per-model `flags`)."""
    import subprocess
    from vtc_amd import _lib as L
    src = tmp_path / "layout.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "vtc_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vtc_block_w), sizeof(vtc_vision_w), sizeof(vtc_text_w), sizeof(vtc_cam_w),
         offsetof(vtc_vision_w, flags), offsetof(vtc_vision_w, pix_mean), offsetof(vtc_vision_w, conv_w), offsetof(vtc_text_w, flags),
         offsetof(vtc_text_w, tok_emb), offsetof(vtc_cam_w, bn_mean));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [L.C.sizeof(L.BlockW), L.C.sizeof(L.VisionW), L.C.sizeof(L.TextW), L.C.sizeof(L.CamW), L.VisionW.flags.offset,
            L.VisionW.pix_mean.offset, L.VisionW.conv_w.offset, L.TextW.flags.offset, L.TextW.tok_emb.offset, L.CamW.bn_mean.offset]
    assert got == want, (got, want)


def test_no_process_wide_switches_in_the_abi():
    """no hidden global state.  The path switches of rounds 1-2 (vtc_set_ln_fold / vtc_set_fused_attention) are
    per-model flags of the weight structs now; the library exports no setter."""
    from vtc_amd import _lib as L
    from vtc_amd import towers
    lib = L.lib()
    assert not any(hasattr(lib, n) for n in ("vtc_set_ln_fold", "vtc_set_fused_attention"))
    assert not hasattr(lib, "vtc_qkv_attention")          # ABI 6: the fused QKV + attention kernel left the product (tools/probes/)
    assert towers.tower_flags() == 0 and towers.tower_flags(ln_fold=False, full_last_layer=True) == 9


def test_every_getenv_of_the_library_is_listed_in_the_header():
    """the library reads environment variables into function-local statics -- process-wide state the ABI's
    signatures do not show.  include/vtc_hip.h lists every one of them; this holds the list to the sources (both ways)."""
    import glob
    hdr = open(os.path.join(ROOT, "include", "vtc_hip.h")).read()
    listed = set(re.findall(r"\b(VTC_(?:GEMM|SWEEP|CAM|PATCH)_[A-Z0-9_]+)=", hdr)) | {"VTC_CAM_TEST_GAVE_UP"}
    used = set()
    for f in glob.glob(os.path.join(ROOT, "vtc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vtc_amd", "csrc", "*.h")):
        used |= set(re.findall(r'getenv\("(VTC_[A-Z0-9_]+)"\)', open(f).read()))
    assert used == listed, (sorted(used - listed), sorted(listed - used))
    # ... and the test hook is compiled into the test build only
    cam = open(os.path.join(ROOT, "vtc_amd", "csrc", "cam.hip")).read()
    i = cam.index('getenv("VTC_CAM_TEST_GAVE_UP")')
    assert "#ifdef VTC_TEST_HOOKS" in cam[max(0, i - 600):i] and "#endif" in cam[i:i + 200]


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
        m = cls(model_type=cfg, **({"n_heads": 2} if kind.endswith("finaltf") else {}))      # TINY features are 128-d: 2 heads of 64
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
    m = HM.PretrainedCLIP_finaltf(model_type=ClipConfig(**asdict(A.TINY)), n_heads=2)
    for blk in m.final_transformer.resblocks:                      # model/model.py:440-450
        assert blk.mlp.c_proj.weight.abs().sum() == 0 and blk.attn.out_proj.weight.abs().sum() == 0
    assert m.final_linear.weight.abs().sum() == 0                  # :452


BASELINE_CONFIGS = ["pretrained_clip.jsonc", "pretrained_clip_comments_attention.jsonc",
                    "pretrained_clip_timesformer_comments_attention.jsonc"]


def _config_paths(sub=""):
    d = os.path.join(ROOT, "configs", sub)
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".jsonc"))


def test_reference_configs_load(tmp_path):
    from vtc_amd.host.parse_config import ConfigParser, read_jsonc
    paths = _config_paths() + _config_paths("synthetic")
    assert len(paths) >= 7
    for p in paths:
        cfg = read_jsonc(p)
        assert "arch" in cfg and "type" in cfg["arch"]
    c = ConfigParser.from_file(paths[0], modification={"arch;args;branch_to_adapt_val": "skip", "batch_size": 7})
    assert c["batch_size"] == 7


def test_shipped_configs_are_the_references_own_files():
    """configs/*.jsonc are the reference's files byte for byte (checked where the reference checkout exists)."""
    ref = "/root/reference/configs"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present on this box")
    names = sorted(f for f in os.listdir(ref) if f.endswith(".jsonc"))
    assert set(BASELINE_CONFIGS) <= set(names)
    for f in names:
        assert open(os.path.join(ref, f), "rb").read() == open(os.path.join(ROOT, "configs", f), "rb").read(), f


@pytest.mark.parametrize("name", BASELINE_CONFIGS + ["pretrained_clip_comments_attn_frozen.jsonc",
                                                     "pretrained_clip_avg_comments.jsonc",
                                                     "pretrained_clip_1frame_comments_attention.jsonc"])
def test_unmodified_reference_configs_resolve_dataset_and_arch(name):
    """evaluation/eval.py:58,88 on the reference's unmodified configs: ``init_obj("dataset", module_data, train=False,
    test=True)`` resolves ImTextDataset / VideoDatasetSegments (empty csv_file -> synthetic pairs with the reference's
    tensor contract) and ``arch.type`` names a class of the drop-in ``model.model`` whose ctor takes ``arch.args``."""
    import inspect as _inspect

    import model.model as module_arch
    from vtc_amd.host import datasets as module_data
    from vtc_amd.host.parse_config import ConfigParser
    cfg = ConfigParser.from_file(os.path.join(ROOT, "configs", name), modification={"dataset;args;n_pairs": 5})
    for train, test in ((False, True), (True, False), (False, False)):
        ds = cfg.init_obj("dataset", module_data, train=train, test=test)
        assert len(ds) == 5
        vis, title, comments, meta = ds[3]
        video = cfg["dataset"]["type"] == "VideoDatasetSegments" and not cfg["dataset"]["args"].get("first_frame_only")
        assert tuple(vis.shape) == ((8, 3, 224, 224) if video else (3, 224, 224)) and vis.dtype == torch.float32
        assert tuple(title.shape) == (77,) and title.dtype == torch.int64 and title[0] == 49406 and title.max() == 49407
        add = {"always": True, "train_only": train, "never": False}[cfg["dataset"]["args"]["add_comments"]]
        nc = int(cfg["dataset"]["args"].get("num_comms", 0)) if add else 0
        assert tuple(comments.shape) == (max(nc, 1), 77) and comments.dtype == torch.int64
        if nc == 0:                                       # _tokenise([""]): one empty comment
            assert comments[0, 0] == 49406 and comments[0, 1] == 49407 and comments[0, 2:].sum() == 0
        assert set(meta) == {"id"}
    cls = getattr(module_arch, cfg["arch"]["type"])
    params = _inspect.signature(cls.__init__).parameters
    assert all(k in params for k in cfg["arch"]["args"]), (name, cfg["arch"]["args"])


def test_real_csv_is_refused_not_faked():
    from vtc_amd.host import datasets as module_data
    with pytest.raises(NotImplementedError):
        module_data.ImTextDataset(csv_file="posts.csv", root="/data")


def test_timesformer_modules_export_the_tower_class():
    """model.timesformer_clip{,_alt}: VisualTransformer with the reference's ctor signature
    (model/timesformer_clip_alt.py:203-213) and a real forward (raises off-GPU instead of falling back)."""
    import model.timesformer_clip as v1
    import model.timesformer_clip_alt as alt
    for mod in (alt, v1):
        got = list(inspect.signature(mod.VisualTransformer.__init__).parameters)[1:]
        assert got == ["input_resolution", "patch_size", "width", "layers", "heads", "output_dim", "nframes"]
        t = mod.VisualTransformer(64, 16, 128, 2, 2, 64, 4).eval()
        assert t.temporal_embed.shape == (4, 128)
        with pytest.raises(RuntimeError, match="GPU"):
            t(torch.zeros(1, 4, 3, 64, 64))
    keys = set(alt.VisualTransformer(64, 16, 128, 1, 2, 64, 4).state_dict())
    assert {"transformer.resblocks.0.temporal_fc.weight", "transformer.resblocks.0.timeattn.in_proj_weight",
            "transformer.resblocks.0.ln_time.weight", "temporal_embed", "proj", "conv1.weight"} <= keys
    assert not any("temporal_fc" in k for k in v1.VisualTransformer(64, 16, 128, 1, 2, 64, 4).state_dict())


def test_fold_ln_packing_identity_fp64():
    """Host logic of the folded LayerNorm (vtc_amd/towers.py::_fold_ln, include/vtc_hip.h vtc_block_w *_wf/_s/_c):
    LN(x) W^T + b == rstd (x W'^T - mean s) + c with W' = gamma . W, s = row sums of W', c = b + W beta -- checked in fp64
    on the packed tensors (fp32 'operand format' so that no rounding enters), shifted rows included (LayerNorm does not
    see a per-row constant: the stream may be stored centred)."""
    import torch
    from vtc_amd import towers

    class Keep:
        def __init__(self):
            self.t = []

        def mat(self, t, dtype):
            self.t.append(t.detach().to(dtype).contiguous())
            return len(self.t) - 1

        def f32(self, t):
            return self.mat(t, torch.float32)

    g = torch.Generator().manual_seed(0)
    W, N, M = 64, 48, 10
    w, b = torch.randn(N, W, generator=g, dtype=torch.float64), torch.randn(N, generator=g, dtype=torch.float64)
    gamma, beta = torch.randn(W, generator=g, dtype=torch.float64), torch.randn(W, generator=g, dtype=torch.float64)
    x = torch.randn(M, W, generator=g, dtype=torch.float64) * 3 + 5
    k = Keep()
    iw, i_s, ic = towers._fold_ln(k, w, b, gamma, beta, torch.float64)
    wf, s, c = k.t[iw].double(), k.t[i_s].double(), k.t[ic].double()
    ref = torch.nn.functional.layer_norm(x, (W,), gamma, beta, 1e-5) @ w.t() + b
    for shift in (0.0, -x.mean(1, keepdim=True)):
        xs = x + shift
        mean, var = xs.mean(1, keepdim=True), xs.var(1, unbiased=False, keepdim=True)
        got = (xs @ wf.t() - mean * s[None]) / torch.sqrt(var + 1e-5) + c[None]
        assert (got - ref).abs().max() < 1e-5          # s, c are stored fp32


def test_packed_weight_signature_tracks_updates_and_reassignment():
    """host/model.py::_signature caches the list of tensors it hashes (one module walk instead of one per forward): in-place
    updates, dtype/device moves and RE-ASSIGNED parameters (global registration hooks) must all change it."""
    import warnings
    import torch
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig  # noqa: F401
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text").eval()
    s0 = m._signature()
    assert m._signature() == s0
    with torch.no_grad():
        m.mask_embedding.add_(1.0)
    s1 = m._signature()
    assert s1 != s0
    m.model.visual.proj = torch.nn.Parameter(m.model.visual.proj.detach().clone())
    s2 = m._signature()
    assert s2 != s1 and m.model.visual.proj.data_ptr() in {p for p, _ in s2[:-2]}
    m.double()
    s3 = m._signature()
    assert s3 != s2
    # a direct write into the registration dict fires no hook (torch.__future__ overwrite-on-conversion, parametrize, pruning)
    m.model.visual._parameters["proj"] = torch.nn.Parameter(m.model.visual.proj.detach().clone())
    s4 = m._signature()
    assert s4 != s3 and m.model.visual.proj.data_ptr() in {p for p, _ in s4[:-2]}


def test_build_refuses_probe_macros():
    """one stray -D must not ship a diagnostic (or, formerly, a wrong-answer) library: build() refuses VTC_* macros
    from the environment, the sources #error on the removed timing probes, and the product Makefile defines none."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, CXXFLAGS="-DVTC_GEMM_STAMPS"))
    assert r.returncode != 0 and "refusing to build" in r.stderr
    src = open(os.path.join(ROOT, "vtc_amd", "csrc", "gemm.hip")).read()
    assert "#error" in src and "VTC_PROBE_MFMA32" in src.split("#error")[0][-800:]       # named only in the guard
    for probe in ("VTC_ABLATE_STORES", "VTC_PROBE_MFMA32", "VTC_ABLATE_DMA", "VTC_GEMM_EXP", "VTC_QKVA_SKIP"):
        for f in ("gemm.hip", "gemm_common.h"):
            body = open(os.path.join(ROOT, "vtc_amd", "csrc", f)).read()
            uses = [ln for ln in body.splitlines() if probe in ln and "defined(" not in ln and not ln.lstrip().startswith("//")]
            assert not uses, (f, probe, uses[:2])


def test_every_advertised_model_type_constructs_or_refuses_at_construction():
    """VERDICT r3 missing #4: `CONFIGS` advertises what the reference's factory does (model/timesformer_clip_alt.py:290-310).  Every
    key builds the wrappers with the reference's state-dict shapes and the reference's DEFAULT n_heads = 8 (ViT-L/14: head_dim 96, the
    generic short-sequence attention core); what the HIP path does not cover (a CAM head_dim beyond 128 or not an integer) says so
    in the constructor, not at the first forward.
    (Forwards of every key: tests/test_gpu_towers.py::test_every_model_type_forward_vs_oracle.)"""
    import warnings
    from dataclasses import replace
    from vtc_amd.host import clip_arch
    from vtc_amd.host import model as HM
    for name, cfg in clip_arch.CONFIGS.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            c = HM.PretrainedCLIP_TimeSformer_finaltf(model_type=name)          # the reference's default n_heads = 8
            P = (cfg.image_resolution // cfg.vision_patch_size) ** 2
            v = c.model.visual
            assert v.positional_embedding.shape == (1 + P, cfg.vision_width) and v.temporal_embed.shape == (8, cfg.vision_width)
            assert v.conv1.weight.shape == (cfg.vision_width, 3, cfg.vision_patch_size, cfg.vision_patch_size)
            assert len(v.transformer.resblocks) == cfg.vision_layers and v.proj.shape == (cfg.vision_width, cfg.embed_dim)
            assert c.final_transformer.width == cfg.embed_dim and c.mask_embedding.shape == (1, cfg.embed_dim)
            del c
    # the refusal, on a small architecture with ViT-L/14's feature width: 768 / 4 = 192 > 128, 768 / 7 not an integer
    small = replace(clip_arch.CONFIGS["ViT-L/14"], vision_layers=1, vision_width=128, transformer_layers=1,
                    vocab_size=49408)        # feature_dim = the text width 768 (model/model.py:394)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for bad in (4, 7):
            with pytest.raises(NotImplementedError, match="head_dim"):
                HM.PretrainedCLIP_finaltf(model_type=small, n_heads=bad)
        HM.PretrainedCLIP_finaltf(model_type=small)                  # default 8 heads of 96
        HM.PretrainedCLIP_finaltf(model_type=small, n_heads=12)
