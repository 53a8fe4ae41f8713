"""GPU: the reference's OWN test file, tests/test_pretrained_clip.py, on the drop-in -- same constructors (`module_arch.PretrainedCLIP()`,
`PretrainedCLIP_finaltf(branch_to_adapt_val=...)` with their ViT-B/32 defaults), same calls, same assertions, line for line.  What differs
is what this container cannot have: `clip.load` downloads the ViT-B/32 checkpoint, so here it reads a synthetic upstream-format state dict
through `VTC_CLIP_WEIGHTS` (fp16 tensors + the three scalar keys, as upstream saves them) -- every model of a test is then built on the SAME
CLIP weights, as with the real checkpoint -- and `clip.tokenize` (the un-vendored BPE) is replaced by seeded token rows of the same layout.
fp32 arithmetic (the reference's, model/model.py:318)."""
import numpy as np
import pytest
import torch

from oracle import arch as A

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture()
def clip_checkpoint(tmp_path, monkeypatch):
    a = A.VIT_B32
    sd = {}
    sd.update(A.synth_visual(a, 401, prefix="visual."))
    sd.update(A.synth_text(a, 402, prefix=""))
    sd = {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}
    sd.update(input_resolution=torch.tensor(224), context_length=torch.tensor(77), vocab_size=torch.tensor(49408))
    path = tmp_path / "ViT-B-32.pt"
    torch.save(sd, path)
    monkeypatch.setenv("VTC_CLIP_WEIGHTS", str(path))
    monkeypatch.setenv("VTC_COMPUTE_DTYPE", "f32")
    return a


def tokenize(n, seed):          # stands in for clip.tokenize([...]) -> [n, 77] int64
    return A.synth_tokens(n, A.VIT_B32, seed)


def test_official_clip_same_as_ours(clip_checkpoint):
    import model.model as module_arch                 # the reference's import path (shim -> vtc_amd.host.model)
    from vtc_amd.host import clip_arch as clip
    ims = torch.randn(2, 3, 224, 224).cuda()
    title = tokenize(2, 1).cuda()
    comms = torch.stack([tokenize(2, 2), tokenize(2, 3)]).cuda()

    clipmodel = clip.load("ViT-B/32", device="cpu").cuda()
    clipmodel.compute_dtype = torch.float32
    clipmodel.visual.compute_dtype = torch.float32

    ourmodel = module_arch.PretrainedCLIP().cuda()
    ourmodel.eval()

    ourmodel_finaltf_skip = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val="skip")
    torch.nn.init.normal_(ourmodel_finaltf_skip.final_linear.weight)
    ourmodel_finaltf_skip = ourmodel_finaltf_skip.cuda()
    ourmodel_finaltf_skip.eval()

    clip_im = clipmodel.encode_image(ims)
    clip_txt = clipmodel.encode_text(title)

    our_im, our_txt, _ = ourmodel(ims, title)

    our_im_skiptf, our_txt_skiptf, _ = ourmodel_finaltf_skip(ims, title, comms)

    # Check equal to off-the-shelf clip
    assert torch.allclose(our_im, clip_im / clip_im.norm(dim=-1, keepdim=True))
    assert torch.allclose(our_txt, clip_txt / clip_txt.norm(dim=-1, keepdim=True))

    # Check version with final transformer is identical
    # when final transformer is skipped
    assert torch.allclose(our_im_skiptf, our_im)
    assert torch.allclose(our_txt_skiptf, our_txt)


def test_branch_to_adapt(clip_checkpoint):
    import model.model as module_arch
    torch.manual_seed(123)

    ims = torch.randn(2, 3, 224, 224).cuda()
    title = tokenize(2, 1).cuda()
    title2 = torch.cat([tokenize(1, 9), tokenize(2, 1)[1:]]).cuda()      # ["goodbye", "world"]: the first title changes
    comms = torch.stack([tokenize(2, 2), tokenize(2, 3)]).cuda()

    m_skip = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val="skip")
    torch.nn.init.normal_(m_skip.final_linear.weight)
    m_vis = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val="image")
    torch.nn.init.normal_(m_vis.final_linear.weight)
    m_txt = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val="text")
    torch.nn.init.normal_(m_txt.final_linear.weight)
    m_skip, m_vis, m_txt = m_skip.cuda().eval(), m_vis.cuda().eval(), m_txt.cuda().eval()

    imf, titlef, _ = m_skip(ims, title, comms)
    imv, titlev, _ = m_vis(ims, title, comms)
    imt, titlet, _ = m_txt(ims, title, comms)

    # Only the adapted modality should change
    assert torch.allclose(imf, imt)
    assert torch.allclose(titlef, titlev)

    assert not torch.allclose(imv, imf)
    assert not torch.allclose(titlet, titlef)

    # Image feat should stay the same when
    # changing title
    imv2, titlev2, _ = m_vis(ims, title2, comms)

    assert torch.allclose(imv2, imv)
    assert not torch.allclose(titlev2, titlev)
