"""GPU: the chunked long-video evaluation path (evaluation/retrieval_evaluation.py:174-264) on ragged
batches vs the oracle running the reference's per-video batch-1 loop."""
import os
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
    # ranks: the table is computed from the GPU's embeddings, so hold it to the fp64 ranks of THOSE embeddings without a
    # near-tie guard (the EXACT sweep's contract), and report how the oracle's own embeddings rank (end-to-end view)
    tvr, vtr = E.compute_recall_table(v_emb.cpu(), c_emb.cpu(), np.float64)
    np.testing.assert_array_equal(table["Video to Text"], tvr)
    np.testing.assert_array_equal(table["Text to Video"], vtr)
    tvr_o, vtr_o = E.compute_recall_table(ref_v, ref_c, np.float64)
    nt = E.near_ties(ref_v.numpy(), ref_c.numpy()) + E.near_ties(ref_c.numpy(), ref_v.numpy())
    print(f"[parity] chunked eval ({branch}): near ties {nt}; oracle-embedding table {tvr_o.tolist()} / {vtr_o.tolist()}")
    if nt == 0:       # 6 videos: a 1e-6 gap between neighbours would be the only way the two embedding sets rank differently
        np.testing.assert_array_equal(table["Video to Text"], tvr_o)
        np.testing.assert_array_equal(table["Text to Video"], vtr_o)


def test_ragged_chunked_eval_vit_b32_bf16_vs_per_video_oracle_loop():
    """VERDICT r3 #9 (f1 was TINY / fp32 only): the real architecture in the headline precision -- ViT-B/32 TimeSformer, bf16
    operands -- on videos of 1 / 3 chunks (1 / 2 / 3 with --extended; a ragged tail among them), real comments and the dummy ones, against the
    oracle's per-video batch-1 loop (evaluation/retrieval_evaluation.py:136,174-259); tolerance 1e-3 on the mean-of-chunks video
    embedding (a mean of unit vectors: |.| <= 1) and on the unit-norm caption embedding."""
    from vtc_amd.host import model as HM
    from vtc_amd.host import retrieval_evaluation as RE
    a = A.VIT_B32
    sd = A.synth_model(a, 63, "timesformer_finaltf", nframes=8)
    g = torch.Generator().manual_seed(64)
    for k in list(sd):
        if k.endswith("temporal_fc.weight") or (k.startswith("final_transformer.") and (k.endswith("out_proj.weight") or k.endswith("c_proj.weight"))):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
    m = HM.PretrainedCLIP_TimeSformer_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
    m.load_state_dict(sd, strict=True)
    m = m.eval().cuda()
    m.compute_dtype = torch.bfloat16
    videos = []
    import sys
    ext = os.environ.get("VTC_TEST_EXTENDED") == "1" or "--extended" in sys.argv
    for i, nfr in enumerate([16 * 8, 16 * 13, 16 * 17 + 3] if ext else [16 * 8, 16 * 17 + 3]):   # 1, (2,) 3 chunks; resampled tails
        fr = A.synth_pixels((nfr, 3, 224, 224), 400 + i).bfloat16().float()        # bf16-representable pixels: both sides see the same input
        cap = A.synth_tokens(1, a, 500 + i)[0]
        com = A.synth_tokens(5, a, 600 + i, empty_frac=0.3) if i != 1 else None        # real comments / the dummy ones
        videos.append((fr, cap, com))
    table, v_emb, c_emb = RE.retrieval_evaluation(m, videos, device="cuda")
    ref_v, ref_c = [], []
    for fr, cap, com in videos:
        chunks = E.chunk_frames(fr[None], 16, 8)
        comments = com[None] if com is not None else RE.empty_comments(1, 5, a.context_length)
        fv, ft, _ = M.pretrained_clip_timesformer_finaltf(chunks, cap[None], comments, sd, a, "text")
        ref_v.append(fv)
        ref_c.append(ft[0])
    ref_v, ref_c = E.mean_chunks(ref_v), torch.stack(ref_c)
    ev, ec = float((v_emb.cpu() - ref_v).abs().max()), float((c_emb.cpu() - ref_c).abs().max())
    print(f"[parity] chunked eval ViT-B/32 bf16: video (mean of chunks) max err {ev:.3e}, caption max err {ec:.3e} (tol 1e-3)")
    assert ev < 1e-3 and ec < 1e-3
    assert set(table) >= {"Video to Text", "Text to Video"}
