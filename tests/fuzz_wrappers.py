"""Random-shape parity probe: the four wrappers on the HIP path vs the live oracle (TINY architecture), odd batch sizes,
1..7 comments per item, all-empty / no-empty comments, fp32 and bf16.  usage: python tests/fuzz_wrappers.py [n_cases]   (test infrastructure: it runs the oracle, so it lives under tests/)"""
import os
import sys
from dataclasses import asdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import arch as A
from oracle import model_ref as M
from vtc_amd.host import model as HM
from vtc_amd.host.clip_arch import ClipConfig

torch.set_grad_enabled(False)
a = A.TINY


def run_cases(n_cases, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    worst = {}
    for case in range(n_cases):
        kind = ["clip", "clip_finaltf", "timesformer", "timesformer_finaltf"][case % 4]
        B = int(rng.choice([1, 2, 3, 5, 7, 13, 33]))
        nc = int(rng.integers(1, 8))
        branch = ["text", "image", "skip"][int(rng.integers(0, 3))]
        act = [None, "squash", "tanh", "normalize", "sub_mean", "bn"][int(rng.integers(0, 6))]
        empty = float(rng.choice([0.0, 0.3, 1.0]))
        dtype = [torch.float32, torch.bfloat16][int(rng.integers(0, 2))]
        ws = int(rng.integers(1, 1000))
        bn = act in ("sub_mean", "bn")
        sd = A.synth_model(a, ws, kind, nframes=8, bn_stats=bn and kind.endswith("finaltf"))
        shape = (B, 8, 3, a.image_resolution, a.image_resolution) if kind.startswith("timesformer") else (B, 3, a.image_resolution, a.image_resolution)
        vis = A.synth_pixels(shape, ws + 1)
        title = A.synth_tokens(B, a, ws + 2)
        comments = A.synth_tokens(B * nc, a, ws + 3, empty_frac=empty).reshape(B, nc, -1)
        cfg = ClipConfig(**asdict(a))
        if kind == "clip":
            m = HM.PretrainedCLIP(model_type=cfg)
            ref = M.pretrained_clip(vis, title, sd, a, None, None)
            args = (vis, title)
        elif kind == "timesformer":
            m = HM.PretrainedCLIP_TimeSformer(model_type=cfg)
            ref = M.pretrained_clip_timesformer(vis, title, sd, a)
            args = (vis, title)
        else:
            cls = HM.PretrainedCLIP_finaltf if kind == "clip_finaltf" else HM.PretrainedCLIP_TimeSformer_finaltf
            m = cls(model_type=cfg, branch_to_adapt_val=branch, residual_activation=act, n_heads=2)
            fn = M.pretrained_clip_finaltf if kind == "clip_finaltf" else M.pretrained_clip_timesformer_finaltf
            ref = fn(vis, title, comments, sd, a, branch, residual_activation=act, n_heads=2)
            args = (vis, title, comments)
        m.load_state_dict(sd, strict=True)
        m = m.eval().cuda()
        m.compute_dtype = dtype
        out = m(*[x.cuda() for x in args])
        scale = float(np.exp(sd["model.logit_scale"].item()))
        errs = [float((o.cpu() - r).abs().max()) for o, r in zip(out[:2], ref[:2])] + [float((out[2].cpu() - ref[2]).abs().max()) / scale]
        # BASELINE.json: 1e-5 (fp32) / 1e-3 (bf16) on 512-d unit-norm embeddings; TINY's 128-d elements are 2x larger
        tol = 1e-5 if dtype == torch.float32 else 1e-3 * (512 / a.embed_dim) ** 0.5
        ok = max(errs) < tol
        key = str(dtype)
        worst[key] = max(worst.get(key, 0.0), max(errs))
        if verbose:
            print(f"{'ok ' if ok else 'BAD'} {kind:20s} B={B:2d} nc={nc} branch={branch:5s} act={act} empty={empty} {dtype}: "
                  f"vis {errs[0]:.2e} text {errs[1]:.2e} cos {errs[2]:.2e}", flush=True)
        assert ok, (kind, B, nc, branch, act, empty, dtype, errs)
    return worst


if __name__ == "__main__":
    print("worst", run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(os.environ.get("SEED", "0"))))
