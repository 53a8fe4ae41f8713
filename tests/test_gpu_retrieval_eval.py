"""GPU: the chunked long-video evaluation path (evaluation/retrieval_evaluation.py:174-264) on ragged
batches vs the oracle running the reference's per-video batch-1 loop."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import arch as A
from oracle import eval_ref as E
from oracle import model_ref as M

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize("branch", ["text", "image", "skip"])
def test_ragged_chunked_eval_matches_per_video_loop(branch):
    from vtc_amd.host import model as HM
    from vtc_amd.host import retrieval_evaluation as RE
    from vtc_amd.host.clip_arch import ClipConfig
    a = A.TINY
    sd = A.synth_model(a, 61, "timesformer_finaltf", nframes=8)
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type=ClipConfig(**asdict(a)), branch_to_adapt_val=branch, n_heads=2)
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    m.compute_dtype = torch.float32
    rng = np.random.default_rng(0)
    videos = []
    for i, nfr in enumerate([200, 16 * 8, 16 * 13, 50, 16 * 17 + 3, 129]):       # 1..3 chunks, ragged tails
        fr = A.synth_pixels((nfr, 3, a.image_resolution, a.image_resolution), 100 + i)
        cap = A.synth_tokens(1, a, 200 + i)[0]
        com = A.synth_tokens(5, a, 300 + i, empty_frac=0.3) if i % 2 == 0 else None   # real comments or dummies
        videos.append((fr, cap, com))
    table, v_emb, c_emb = RE.retrieval_evaluation(m, videos, device="cuda")
    # oracle: the reference's loop, one video at a time
    ref_v, ref_c = [], []
    for fr, cap, com in videos:
        chunks = E.chunk_frames(fr[None], 16, 8)
        comments = com[None] if com is not None else RE.empty_comments(1, 5, a.context_length)
        ncomm = chunks.shape[0] if branch == "image" else 1
        fv, ft, _ = M.pretrained_clip_timesformer_finaltf(chunks, cap[None], comments.expand(ncomm, -1, -1), sd, a, branch, n_heads=2)
        ref_v.append(fv)
        ref_c.append(ft[0])
    ref_v, ref_c = E.mean_chunks(ref_v), torch.stack(ref_c)
    assert (v_emb.cpu() - ref_v).abs().max() < 1e-5 and (c_emb.cpu() - ref_c).abs().max() < 1e-5
    assert not np.allclose(np.linalg.norm(v_emb.cpu().numpy(), axis=1), 1.0, atol=1e-4)      # mean is NOT renormalised
    tvr, vtr = E.compute_recall_table(ref_v, ref_c)
    if E.near_ties(ref_v.numpy(), ref_c.numpy()) == 0 and E.near_ties(ref_c.numpy(), ref_v.numpy()) == 0:
        np.testing.assert_allclose(table["Video to Text"], tvr)
        np.testing.assert_allclose(table["Text to Video"], vtr)
