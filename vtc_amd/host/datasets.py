"""Synthetic stand-in for the reference's dataset loaders (dataset_loaders/*.py, out of scope: they
need torchvision video IO, PIL, ffmpeg and the VTC csv/images).  Only the TENSOR CONTRACT the hot
path consumes is reproduced (evaluation/eval.py:101-116): items are
``(vis [3,224,224] | [F,3,224,224], title [77] int64, comments [nc,77] int64, meta {"id": int})``.
Pixels ~ N(0,1) (the distribution tests/test_pretrained_clip.py:8 uses); tokens are
``[SOT, t_1..t_L, EOT, 0...]`` with L ~ U{1..75}; 10 % of comments are the empty string."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

SOT, EOT = 49406, 49407


def synth_tokens(n, ctx, gen, empty_frac=0.0):
    out = torch.zeros(n, ctx, dtype=torch.int64)
    lens = torch.randint(1, ctx - 1, (n,), generator=gen)
    lens[torch.rand(n, generator=gen) < empty_frac] = 0
    toks = torch.randint(1, SOT, (n, ctx), generator=gen)
    pos = torch.arange(ctx)[None]
    out = torch.where((pos >= 1) & (pos <= lens[:, None]), toks, out)
    out[:, 0] = SOT
    out[torch.arange(n), lens + 1] = EOT
    return out


class SyntheticPairs(Dataset):
    def __init__(self, n_pairs=1024, kind="image", add_comments="always", num_comms=5, seed=123, nframes=8,
                 resolution=224, context=77, train=False, test=True, **_):
        self.n, self.kind, self.nc, self.seed = int(n_pairs), kind, int(num_comms), int(seed)
        self.nframes, self.res, self.ctx = nframes, resolution, context
        self.add_comments = add_comments
        g = torch.Generator().manual_seed(self.seed)
        self.titles = synth_tokens(self.n, context, g)
        self.comments = synth_tokens(self.n * self.nc, context, g, empty_frac=0.1).reshape(self.n, self.nc, context)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        shape = (3, self.res, self.res) if self.kind == "image" else (self.nframes, 3, self.res, self.res)
        vis = torch.randn(shape, generator=g)
        return vis, self.titles[i], self.comments[i], {"id": i}
