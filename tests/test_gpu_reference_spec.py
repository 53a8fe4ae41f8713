"""GPU: the four properties the reference's own test file pins (tests/test_pretrained_clip.py), stated in this repo's
terms and checked on the drop-in with the oracle as a third opinion:

  P1  a plain wrapper returns the L2-normalised outputs of the CLIP towers it wraps            (reference :36-37)
  P2  a CAM wrapper with ``branch_to_adapt_val="skip"`` returns what the plain wrapper returns  (:41-42)
  P3  adapting one branch changes that branch's embedding and only that one                     (:74-78)
  P4  with the image branch adapted, the image embedding does not depend on the title           (:84-85)

The reference builds every model of a test on the downloaded ViT-B/32 checkpoint; here all models of a case load ONE synthetic
upstream-format state dict through ``VTC_CLIP_WEIGHTS`` (fp16 tensors + the three scalar keys, as upstream saves them), tokens are
seeded rows of ``clip.tokenize``'s layout, and the cases run in the reference's fp32 arithmetic (``torch.allclose`` defaults, as the
reference asserts) and in the 16-bit default (1e-3)."""
import pytest
import torch

from oracle import arch as A
from oracle import clip_ref as CR
from oracle import model_ref as M

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ARCH = A.VIT_B32
PRECISIONS = {"f32": dict(rtol=1e-5, atol=1e-8), "bf16": dict(rtol=0.0, atol=1e-3)}


@pytest.fixture(params=sorted(PRECISIONS))
def setup(request, tmp_path, monkeypatch):
    """(close, upstream state dict, inputs): one CLIP checkpoint for every model of the case."""
    sd = {**A.synth_visual(ARCH, 401, prefix="visual."), **A.synth_text(ARCH, 402, prefix="")}
    saved = {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}
    saved.update(input_resolution=torch.tensor(224), context_length=torch.tensor(77), vocab_size=torch.tensor(49408))
    torch.save(saved, tmp_path / "ViT-B-32.pt")
    monkeypatch.setenv("VTC_CLIP_WEIGHTS", str(tmp_path / "ViT-B-32.pt"))
    monkeypatch.setenv("VTC_COMPUTE_DTYPE", request.param)
    tol = PRECISIONS[request.param]
    g = torch.Generator().manual_seed(123)
    inputs = dict(ims=torch.randn(2, 3, 224, 224, generator=g).cuda(), title=A.synth_tokens(2, ARCH, 1).cuda(),
                  other_title=torch.cat([A.synth_tokens(1, ARCH, 9), A.synth_tokens(2, ARCH, 1)[1:]]).cuda(),      # item 0 re-titled
                  comms=torch.stack([A.synth_tokens(2, ARCH, 2), A.synth_tokens(2, ARCH, 3)]).cuda())
    half_sd = {"model." + k: v.float() for k, v in saved.items() if torch.is_tensor(v) and v.dim() > 0}           # what the models hold
    return (lambda x, y: torch.allclose(x, y, **tol)), half_sd, inputs


def cam_model(branch):
    import model.model as module_arch                 # the reference's import path (shim -> vtc_amd.host.model)
    m = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val=branch)
    torch.nn.init.normal_(m.final_linear.weight)      # (as the reference's tests do; init_from_avg ignores final_linear)
    return m.cuda().eval()


def test_plain_wrapper_is_the_normalised_towers_and_skip_equals_plain(setup):
    import model.model as module_arch
    from vtc_amd.host import clip_arch
    close, sd, x = setup
    towers = clip_arch.load("ViT-B/32", device="cpu").cuda()
    from vtc_amd.host.model import default_compute_dtype
    towers.compute_dtype = towers.visual.compute_dtype = default_compute_dtype()
    plain = module_arch.PretrainedCLIP().cuda().eval()
    im, txt, _ = plain(x["ims"], x["title"])
    t_im, t_txt = towers.encode_image(x["ims"]), towers.encode_text(x["title"])
    assert close(im, t_im / t_im.norm(dim=-1, keepdim=True)) and close(txt, t_txt / t_txt.norm(dim=-1, keepdim=True))      # P1
    im_s, txt_s, _ = cam_model("skip")(x["ims"], x["title"], x["comms"])
    assert close(im_s, im) and close(txt_s, txt)                                                                            # P2
    # third opinion: the oracle on the weights the models actually hold (the fp16-rounded checkpoint)
    o_im = M.normalize(CR.encode_image(x["ims"].cpu(), sd, ARCH, "model.visual."))
    o_txt = M.normalize(CR.encode_text(x["title"].cpu(), sd, ARCH, "model."))
    assert close(im.cpu(), o_im) or (im.cpu() - o_im).abs().max() < 1e-5
    assert close(txt.cpu(), o_txt) or (txt.cpu() - o_txt).abs().max() < 1e-5


def test_only_the_adapted_branch_moves(setup):
    close, _, x = setup
    models = {b: cam_model(b) for b in ("skip", "image", "text")}
    out = {b: m(x["ims"], x["title"], x["comms"]) for b, m in models.items()}
    (im_s, txt_s, _), (im_i, txt_i, _), (im_t, txt_t, _) = out["skip"], out["image"], out["text"]
    assert close(im_s, im_t) and close(txt_s, txt_i)                        # P3: the other branch is untouched ...
    assert not close(im_i, im_s) and not close(txt_t, txt_s)                # ... and the adapted one is not
    im_i2, txt_i2, _ = models["image"](x["ims"], x["other_title"], x["comms"])
    assert close(im_i2, im_i) and not close(txt_i2, txt_i)                  # P4
