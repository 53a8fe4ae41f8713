"""CPU: host-side pieces of the eval entry point that had no test (VERDICT r4 missing #5, #6):
``add_irrelevant_comms`` (evaluation/eval.py:23-47) and the ``VTC_CLIP_WEIGHTS`` loader that stands in for
``clip.load`` (model/model.py:317)."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import arch as A


def test_add_irrelevant_comms_appends_comments_of_other_items():
    """evaluation/eval.py:23-47: every item keeps its own ``nc`` comments in place and gets ``k`` more, each one a
    comment ROW of some item of the batch at the drawn comment index; the numpy draws come in the reference's order
    (``k`` comment indices, then one batch index per comment, re-drawn once on a self-hit), so a seeded run is
    reproducible and the self-hit rate is the chance of two consecutive self draws (1/bs^2)."""
    from vtc_amd.host.eval import add_irrelevant_comms
    bs, nc, ntok, k = 12, 5, 7, 3
    comments = torch.arange(bs * nc * ntok, dtype=torch.int32).reshape(bs, nc, ntok)      # every row is unique
    np.random.seed(5)
    out = add_irrelevant_comms(comments, k)
    assert out.shape == (bs, nc + k, ntok) and out.dtype == torch.int64                     # :46 `.long()`
    assert torch.equal(out[:, :nc], comments.long())
    # replay the draws: comment indices first (:33), then per comment a batch index, re-drawn once when it is i (:36-40)
    np.random.seed(5)
    self_hits = 0
    for i in range(bs):
        comm_indices = np.random.randint(low=0, high=nc, size=k)
        for j, ci in enumerate(comm_indices):
            bi = int(np.random.randint(low=0, high=bs, size=[1])[0])
            if bi == i:
                bi = int(np.random.randint(low=0, high=bs, size=[1])[0])
            self_hits += bi == i
            assert torch.equal(out[i, nc + j], comments[bi, ci].long()), (i, j)
    assert self_hits <= 1
    # k == 0 is the identity; the batch-size precondition is the caller's assert (:105-107)
    assert torch.equal(add_irrelevant_comms(comments, 0), comments.long())
    # rows of the appended part always exist somewhere in the batch at the same comment slot
    flat = {tuple(r.tolist()) for r in comments.reshape(-1, ntok)}
    assert all(tuple(r.tolist()) in flat for r in out[:, nc:].reshape(-1, ntok))


def _upstream_checkpoint(a, seed, dtype=torch.float16):
    """A state dict in the form upstream ``clip.load(..., jit=False)`` models save: visual.* / transformer.* /
    token_embedding / ... keys without a prefix, fp16 tensors (upstream converts weights with convert_weights), and the
    three scalar entries ``input_resolution`` / ``context_length`` / ``vocab_size`` that upstream's build_model deletes."""
    sd = {}
    sd.update(A.synth_visual(a, seed * 7 + 1, prefix="visual."))
    sd.update(A.synth_text(a, seed * 7 + 2, prefix=""))
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    sd["input_resolution"] = torch.tensor(a.image_resolution)
    sd["context_length"] = torch.tensor(a.context_length)
    sd["vocab_size"] = torch.tensor(a.vocab_size)
    return sd


@pytest.mark.parametrize("wrapped", [False, True])
def test_vtc_clip_weights_loader_reads_an_upstream_state_dict(tmp_path, monkeypatch, wrapped):
    from vtc_amd.host import clip_arch as CA
    a = A.TINY
    sd = _upstream_checkpoint(a, 31)
    path = tmp_path / "clip_tiny.pt"
    torch.save({"state_dict": sd} if wrapped else sd, path)
    monkeypatch.setenv("VTC_CLIP_WEIGHTS", str(path))
    assert CA.pretrained_weights_available()
    m = CA.load(CA.ClipConfig(**asdict(a)))
    got = m.state_dict()
    want = {k: v for k, v in sd.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    assert set(got) == set(want)                                   # strict: nothing missing, nothing unexpected
    for k, v in want.items():
        assert got[k].dtype == torch.float32                       # model/model.py:318 `self.model.float()`
        assert torch.equal(got[k], v.float()), k
    assert not m.training
    # a checkpoint of another architecture is refused by the strict load, not silently truncated
    bad = dict(sd)
    bad["visual.proj"] = bad["visual.proj"][:, :-1]
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        CA.load(CA.ClipConfig(**asdict(a)))
    bad = dict(sd)
    bad["visual.surplus.weight"] = torch.zeros(3)
    torch.save(bad, path)
    with pytest.raises(RuntimeError):
        CA.load(CA.ClipConfig(**asdict(a)))


def test_wrapper_built_on_loaded_weights_carries_them(tmp_path, monkeypatch):
    """PretrainedCLIP(model_type) -> clip_arch.load: the wrapper's ``model.*`` entries are the file's tensors, and the
    TimeSformer tower is initialised from the ViT weights with only time / temporal keys left at their init
    (model/timesformer_clip_alt.py:318-328)."""
    from vtc_amd.host import clip_arch as CA
    from vtc_amd.host import model as HM
    a = A.TINY
    sd = _upstream_checkpoint(a, 32)
    path = tmp_path / "clip_tiny.pt"
    torch.save(sd, path)
    monkeypatch.setenv("VTC_CLIP_WEIGHTS", str(path))
    cfg = CA.ClipConfig(**asdict(a))
    m = HM.PretrainedCLIP(model_type=cfg)
    msd = m.state_dict()
    for k, v in sd.items():
        if k in ("input_resolution", "context_length", "vocab_size"):
            continue
        assert torch.equal(msd["model." + k], v.float()), k
    HM.PretrainedCLIP_TimeSformer.nframes = 8
    t = HM.PretrainedCLIP_TimeSformer(model_type=cfg)
    tsd = t.state_dict()
    for k, v in sd.items():
        if k.startswith("visual."):
            assert torch.equal(tsd["model." + k], v.float()), k
    extra = [k for k in tsd if k.startswith("model.visual.") and k[len("model."):] not in sd]
    assert extra and all(("time" in k or "temporal" in k) for k in extra)


def test_compute_dtype_selection(monkeypatch):
    """The reference computes in fp32 (model/model.py:318); the drop-in defaults to 16-bit operands and says so.
    ``VTC_COMPUTE_DTYPE`` / ``evaluation/eval.py --dtype`` select the arithmetic without touching code."""
    from vtc_amd.host import model as HM
    assert HM.parse_compute_dtype("f32") is torch.float32 and HM.parse_compute_dtype("fp32") is torch.float32
    assert HM.parse_compute_dtype("bf16") is torch.bfloat16 and HM.parse_compute_dtype("float32") is torch.float32
    assert HM.parse_compute_dtype(None) is None
    with pytest.raises(ValueError):
        HM.parse_compute_dtype("fp8")
    monkeypatch.delenv("VTC_COMPUTE_DTYPE", raising=False)
    assert HM.default_compute_dtype() is torch.bfloat16
    monkeypatch.setenv("VTC_COMPUTE_DTYPE", "f32")
    assert HM.default_compute_dtype() is torch.float32
    m = HM.PretrainedCLIP(model_type=HM.clip_arch.ClipConfig(**asdict(A.TINY)))
    assert m.compute_dtype is torch.float32            # picked up at construction
    monkeypatch.setenv("VTC_COMPUTE_DTYPE", "bf16")
    m = HM.PretrainedCLIP(model_type=HM.clip_arch.ClipConfig(**asdict(A.TINY)))
    assert m.compute_dtype is torch.bfloat16
