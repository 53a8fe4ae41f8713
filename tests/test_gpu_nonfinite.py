"""GPU: NaN embeddings never pass silently.  Two paths of the build can return NaN rows with rc 0 -- a
text tower whose IEEE-half blocks overflow in a batch AFTER the first (the range guard switches to bf16 at the next call), and a
one-launch CAM whose grid barrier gave up -- so: every wrapper forward ends with a non-finite flag over the embeddings it returns, read at
the model's next forward / check_finite(); the eval entry point and RecallAtK check unconditionally."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import arch as A

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model_with_outlier_token(token, seed=31):
    from vtc_amd.host import model as HM
    torch.manual_seed(seed)
    m = HM.PretrainedCLIP_finaltf(model_type="ViT-B/32", branch_to_adapt_val="text")
    sd = m.state_dict()
    sd["model.token_embedding.weight"][token, 3] = 1.0e5            # beyond the IEEE-half range (+-65504): inf in the half-operand blocks
    m.load_state_dict(sd, strict=True)
    return m, sd


def test_wrapper_forward_watchdog_raises_at_the_next_forward_and_in_check_finite():
    a = A.VIT_B32
    m, _ = _model_with_outlier_token(777)
    m = m.eval().cuda()
    assert m.compute_dtype == torch.bfloat16
    B = 4
    vis = A.synth_pixels((B, 3, 224, 224), 1).cuda()
    clean = A.synth_tokens(B, a, 2)
    clean[clean == 777] = 778
    comments = A.synth_tokens(B * 5, a, 3, empty_frac=0.2)
    comments[comments == 777] = 778
    comments = comments.reshape(B, 5, -1).cuda()
    out = m(vis, clean.cuda(), comments)
    m.check_finite()                                                   # clean batch: nothing to report
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
    bad = clean.clone()
    bad[2, 2] = 777                                                    # the overflowing token, in a LATER batch than the first
    out = m(vis, bad.cuda(), comments)                                 # returns NaN rows with rc 0 ...
    with pytest.raises(RuntimeError, match="non-finite values in its text embeddings"):
        m.check_finite()                                               # ... and the watchdog says so at the caller's next synchronisation
    assert not torch.isfinite(out[1][2]).all() and torch.isfinite(out[1][:2]).all()
    # without check_finite: the NEXT forward raises once the flag has landed in pinned memory (here: after a sync), then the model works
    # again with the text blocks re-packed as bf16
    out = m(vis, clean.cuda(), comments)        # (switches the tower: warning)
    m.check_finite()
    m2, _ = _model_with_outlier_token(777)
    m2 = m2.eval().cuda()
    m2(vis, clean.cuda(), comments)
    m2(vis, bad.cuda(), comments)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="an earlier forward returned non-finite"):
        m2(vis, clean.cuda(), comments)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        again = m2(vis, bad.cuda(), comments)                           # flag cleared, tower now bf16: finite
    m2.check_finite()
    assert torch.isfinite(again[1]).all()


def test_forced_half_overflow_in_batch_3_of_an_eval_run_raises_instead_of_writing_a_json(tmp_path):
    """evaluation/eval.py on config 2 with a checkpoint whose token_embedding holds one out-of-range value, for a token that first occurs
    in the THIRD batch: the first forward's synchronous range check passes, batch 3 returns NaN rows -- the run must raise and must not
    write the result JSON."""
    from vtc_amd.host import datasets as D
    from vtc_amd.host import eval as ev
    cfg = os.path.join(ROOT, "configs", "pretrained_clip_comments_attention.jsonc")
    n, bs = 96, 16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds = D.ImTextDataset("", train=False, test=True, add_comments="always", num_comms=5, n_pairs=n)
    toks = [torch.cat([ds.titles[i * bs:(i + 1) * bs].reshape(-1), ds.comments[i * bs:(i + 1) * bs].reshape(-1)]) for i in range(n // bs)]
    early = set(torch.cat(toks[:2]).tolist())
    token = next(t for t in toks[2].tolist() if t not in early and 0 < t < 49406)
    m, sd = _model_with_outlier_token(token)
    ckpt = tmp_path / "model.pth"
    torch.save({"state_dict": sd, "config": {"arch": {"args": {}}}}, ckpt)
    with pytest.raises(RuntimeError, match="non-finite"):
        ev.cli(["-c", cfg, "-r", str(ckpt), "--bs", str(bs), "--n_pairs", str(n), "--nc", "5", "--ac", "always"])
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".json")]
    # the same run in fp32 arithmetic (the reference's) has no half blocks: it completes and writes the file
    out, _, _ = ev.cli(["-c", cfg, "-r", str(ckpt), "--bs", str(bs), "--n_pairs", str(n), "--nc", "5", "--ac", "always", "--dtype", "f32"])
    assert [f for f in os.listdir(tmp_path) if f.endswith(".json")] and all(np.isfinite(v) for k, v in out.items() if k.startswith("R"))
