"""GPU parity of the adapter-only training step against the oracle (oracle/train_ref.py, itself
pinned by the reference's train-mode run in tests/golden/train_step_tiny.npz) and against that golden directly."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import arch as A
from oracle import clip_ref as CR
from oracle import train_ref as TR

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def test_train_step_matches_reference_golden():
    """Same inputs, seeds and draws as the reference's own two training steps (golden): loss, every adapter gradient,
    parameters after two Adam(amsgrad) steps."""
    from vtc_amd.host.adapter_train import AdapterTrainer
    case, g = load_golden("train_step_tiny.npz")
    a = A.TINY
    sd = A.synth_model(a, case["wseed"], "clip_finaltf")
    B = case["B"]
    vis = A.synth_pixels((B, 3, a.image_resolution, a.image_resolution), case["xseed"])
    title = A.synth_tokens(B, a, case["tseed"])
    comments = A.synth_tokens(B * 5, a, case["cseed"], empty_frac=case["empty_frac"]).reshape(B, 5, -1)
    fv = CR.encode_image(vis, sd, a, "model.visual.").float()          # frozen towers (constants of the step)
    ft = CR.encode_text(title, sd, a, "model.").float()
    fc = CR.encode_text(comments.reshape(B * 5, -1), sd, a, "model.").float().reshape(B, 5, -1).permute(1, 0, 2).contiguous()
    empty = comments[..., 1] == A.EOT
    tr = AdapterTrainer({k: v.cuda() for k, v in sd.items()}, n_heads=case["n_heads"], lr=case["lr"])
    for step, seed in enumerate(case["rng_seeds"]):
        torch.manual_seed(seed)
        torch.rand([])
        skip = torch.rand(B) > 0.5
        loss = float(tr.step(fv.cuda(), ft.cuda(), fc.cuda(), empty.cuda(), skip.cuda()).cpu())
        assert abs(loss - float(g[f"loss{step}"])) < 5e-6 * max(1.0, abs(loss)), (step, loss, float(g[f"loss{step}"]))
        if step == 0:
            for k, v in tr.grads.items():
                assert _rel(v.cpu().numpy(), g["grad0:" + k]) < 2e-4, (k, _rel(v.cpu().numpy(), g["grad0:" + k]))
    for k, p in tr.params.items():
        d = np.abs(p.cpu().numpy() - g["after2:" + k])
        big = np.abs(g["grad0:" + k]) > 1e-4 * np.abs(g["grad0:" + k]).max()
        assert d[big].max(initial=0.0) < 2e-5, (k, d[big].max())          # Adam amplifies relative gradient error near g ~ eps
        assert d.max() <= 2.001 * case["lr"], k


@pytest.mark.parametrize("branch", ["text", "image"])
def test_train_step_full_width_vs_oracle(branch):
    """Real CAM size (width 512, 8 heads, 2 layers), B = 40 (not a multiple of the padding granule), 5 comments with
    empty ones, random skip draw: three steps against the oracle's autograd + restated Adam."""
    from vtc_amd.host.adapter_train import AdapterTrainer
    a = A.VIT_B32
    gen = torch.Generator().manual_seed(5)
    sd = {k: v for k, v in A.synth_model(a, 61, "clip_finaltf").items()
          if k.startswith("final_transformer.") or k in ("mask_embedding", "model.logit_scale", "final_linear.weight")}
    B, nc, D = 40, 5, 512
    fv, ft = torch.randn(B, D, generator=gen), torch.randn(B, D, generator=gen)
    fc = torch.randn(nc, B, D, generator=gen)
    empty = torch.rand(B, nc, generator=gen) < 0.3
    tr = AdapterTrainer({k: v.cuda() for k, v in sd.items()}, branch=branch)
    osd = {k: v.clone() for k, v in sd.items()}
    opt = TR.AdamAmsgrad({k: osd[k] for k in TR.adapter_param_names(osd)})
    for step in range(3):
        skip = torch.rand(B, generator=gen) > 0.5
        loss = float(tr.step(fv.cuda(), ft.cuda(), fc.cuda(), empty.cuda(), skip.cuda()).cpu())
        ref_loss, ref_g = TR.train_step(fv, ft, fc, empty, skip, osd, opt, branch=branch)
        assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
        for k, v in ref_g.items():
            assert _rel(tr.grads[k].cpu().numpy(), v.numpy()) < 5e-4, (step, k, _rel(tr.grads[k].cpu().numpy(), v.numpy()))
    assert loss < 10.0 and np.isfinite(loss)
