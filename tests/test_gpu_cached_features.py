"""GPU: (f2) cached-feature producer + the 2-D precomputed-feature fast path (model.py:328-330,460-462),
(f3) raw-uint8 pixels with ToTensor+Normalize fused into the patch gather."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import arch as A
from oracle import clip_ref as CR
from oracle import model_ref as M

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
MEAN = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
STD = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)


def test_uint8_pixels_fused_normalise():
    from vtc_amd import towers
    a = A.TINY
    sd = A.synth_visual(a, 71, prefix="v.")
    u8 = torch.randint(0, 256, (5, 3, a.image_resolution, a.image_resolution), dtype=torch.uint8)
    ref = CR.encode_image((u8.float() / 255.0 - MEAN) / STD, sd, a, "v.").numpy()
    pv = towers.PackedVision({k: v.cuda() for k, v in sd.items()}, "v.", torch.float32)
    out = pv.forward(u8.cuda()).cpu().numpy()
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_uint8_frames_at_full_size_through_the_patch_gather():
    """f3 at the real size: raw uint8 frames -- what the loader holds before CLIP_TRANSFORM's ToTensor + Normalize
    (dataset_loaders/dataset_loaders.py:40-49) -- into the ViT-B/32 8-frame TimeSformer and the image tower in the 16-bit mode:
    ONE pass normalises them into the operand format and the patch GEMM gathers from that tensor by LDS-DMA (no im2row matrix;
    towers.hip).  Against the fp32 oracle fed with the normalised float pixels, at the 16-bit tolerance; fp32 mode (im2row path)
    at 1e-5.  An odd batch exercises the tail row tile."""
    from vtc_amd import towers
    from oracle import timesformer_ref as T
    a = A.VIT_B32
    unit = lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True)   # noqa: E731
    g = torch.Generator().manual_seed(5)
    sdv = A.synth_visual(a, 171, nframes=8, prefix="v.")
    for k in list(sdv):
        if k.endswith("temporal_fc.weight"):
            sdv[k] = torch.randn(sdv[k].shape, generator=g) * 0.02
    u8 = torch.randint(0, 256, (2, 8, 3, 224, 224), dtype=torch.uint8, generator=g)
    x = (u8.float() / 255.0 - MEAN[None]) / STD[None]
    ref = unit(T.timesformer_alt(x, sdv, a, "v.").numpy())
    cuda_sd = {k: v.cuda() for k, v in sdv.items()}
    got16 = unit(towers.PackedVision(cuda_sd, "v.", torch.bfloat16).forward(u8.cuda()).cpu().numpy())
    got32 = unit(towers.PackedVision(cuda_sd, "v.", torch.float32).forward(u8.cuda()).cpu().numpy())
    print(f"[parity] uint8 video frames: bf16 {np.abs(got16 - ref).max():.3e}  fp32 {np.abs(got32 - ref).max():.3e}")
    assert np.abs(got16 - ref).max() < 1e-3 and np.abs(got32 - ref).max() < 1e-5
    # the same frames handed over already normalised, in bf16: the gather reads them in place -- same embedding within a
    # fraction of the tolerance (the two normalisations round a few pixels differently)
    alt = unit(towers.PackedVision(cuda_sd, "v.", torch.bfloat16).forward(x.bfloat16().cuda()).cpu().numpy())
    assert np.abs(alt - got16).max() < 5e-4
    sdi = A.synth_visual(a, 172, prefix="v.")
    img = torch.randint(0, 256, (13, 3, 224, 224), dtype=torch.uint8, generator=g)
    refi = unit(CR.encode_image((img.float() / 255.0 - MEAN) / STD, sdi, a, "v.").numpy())
    goti = unit(towers.PackedVision({k: v.cuda() for k, v in sdi.items()}, "v.", torch.bfloat16).forward(img.cuda()).cpu().numpy())
    assert np.abs(goti - refi).max() < 1e-3


def test_cached_features_round_trip_and_2d_fast_path(tmp_path):
    from vtc_amd.host import cache_features as CF
    from vtc_amd.host import model as HM
    from vtc_amd.host.clip_arch import ClipConfig
    a = A.TINY
    sd = A.synth_model(a, 72, "clip_finaltf")
    m = HM.PretrainedCLIP_finaltf(model_type=ClipConfig(**asdict(a)), branch_to_adapt_val="text", n_heads=2)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    m.compute_dtype = torch.float32
    N = 7
    imgs = A.synth_pixels((N, 3, a.image_resolution, a.image_resolution), 73)
    ids = [900 + 3 * i for i in range(N)]
    path = str(tmp_path / "clip_vit_embeddings.pth")
    saved = CF.cache_clip_vit_embeddings(m, [imgs[:4], imgs[4:]], ids, path)
    assert saved["reddit_ids"].dtype is torch.int64 and saved["embeddings"].dtype is torch.float32
    ref = CR.encode_image(imgs, sd, a, "model.visual.")
    assert (saved["embeddings"] - ref).abs().max() < 2e-5 * ref.abs().max()
    order = [ids[i] for i in (3, 0, 6, 5)]
    feats = CF.load_features(order, path)
    title = A.synth_tokens(4, a, 74)
    comments = A.synth_tokens(20, a, 75, empty_frac=0.3).reshape(4, 5, -1)
    out2d = m(feats.cuda(), title.cuda(), comments.cuda())                       # 2-D: precomputed feature
    out4d = m(imgs[[3, 0, 6, 5]].cuda(), title.cuda(), comments.cuda())
    for x, y in zip(out2d, out4d):
        assert (x - y).abs().max() < 1e-5 * max(1.0, float(y.abs().max()))
    r = M.pretrained_clip_finaltf(feats, title, comments, sd, a, "text", n_heads=2)
    assert (out2d[1].cpu() - r[1]).abs().max() < 1e-5
