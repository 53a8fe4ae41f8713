"""Stand-ins for the reference's dataset loaders (dataset_loaders/dataset_loaders.py, out of scope: they
need torchvision video IO, PIL, ffmpeg, the CLIP tokenizer, RAKE and the VTC csv/images).  Only the TENSOR
CONTRACT the hot path consumes is reproduced (evaluation/eval.py:101-116): items are
``(vis [3,224,224] | [F,3,224,224], title [77] int64, comments [nc,77] int64, meta {"id": int})``.

``ImTextDataset`` (dataset_loaders.py:924-1046) and ``VideoDatasetSegments`` (:440-566) keep the
reference's class names and constructor keywords, so that ``config.init_obj("dataset", module_data,
train=..., test=...)`` (evaluation/eval.py:58, train.py:45-64) resolves the reference's UNMODIFIED
``configs/pretrained_clip*.jsonc``.  Those configs ship with ``"csv_file": ""`` (the VTC data is not
released, README.md:105): an empty ``csv_file`` selects synthetic pairs of the same tensor contract;
a non-empty one raises -- reading real posts needs the loaders above.

Synthetic items: pixels ~ N(0,1) (the distribution tests/test_pretrained_clip.py:8 uses); tokens are
``[SOT, t_1..t_L, EOT, 0...]`` with L ~ U{1..75}; 10 % of comments are the empty string
``[SOT, EOT, 0...]`` (what ``_tokenise([""])`` produces, dataset_loaders.py:224-248, and what
``preprocess_comments`` pads with, :274-275)."""
from __future__ import annotations

import os

import torch
from torch.utils.data import Dataset

SOT, EOT = 49406, 49407


def pack_token_lists(token_lists, max_len=77, device=None, summarise=None, strict=False):
    """The array-building half of the reference's `_tokenise` (dataset_loaders/dataset_loaders.py:224-248) on the GPU: one list of
    BPE ids per text (the encoder's output; the encoder itself and the RAKE summariser stay host text processing) -> ids
    [n, max_len] int64 on `device`, as `PretrainedCLIP*.forward` takes them.  One ragged H2D copy + one kernel.

    Over-long texts (`len(ids) + 2 >= max_len`): the reference first re-encodes the RAKE keyword summary of the text and truncates
    only if that is still too long (:235-243).  `summarise(i) -> ids` is the caller's hook for that step (RAKE + BPE of text i);
    without one the ORIGINAL ids are truncated, which differs from the reference for such texts -- a warning says so (`strict`:
    an exception instead)."""
    from .. import ops
    device = torch.device(device if device is not None else "cuda")
    over = [i for i, t in enumerate(token_lists) if len(t) + 2 >= max_len]
    if over and summarise is not None:
        token_lists = list(token_lists)
        for i in over:
            token_lists[i] = list(summarise(i))
    elif over:
        msg = (f"pack_token_lists: {len(over)} text(s) reach max_len={max_len} (first: index {over[0]}); the reference summarises such "
               "texts with RAKE before truncating (dataset_loaders.py:235-243) -- pass summarise=, or the ids differ from the reference's")
        if strict:
            raise ValueError(msg)
        import warnings
        warnings.warn(msg, stacklevel=2)
    lens = torch.tensor([len(t) for t in token_lists], dtype=torch.int32)
    offsets = torch.zeros(len(token_lists) + 1, dtype=torch.int32)
    offsets[1:] = torch.cumsum(lens, 0)
    flat = torch.tensor([int(v) for t in token_lists for v in t] or [0], dtype=torch.int32)
    return ops.pack_tokens(flat.to(device), offsets.to(device), max_len, SOT, EOT)


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


def _should_add_comments(add_comments, train):
    """dataset_loaders.py:215-222."""
    cases = {"always": [True, True], "train_only": [False, True], "never": [False, False]}
    return cases[add_comments][int(bool(train))]


class _ReferenceNamedDataset(SyntheticPairs):
    """Shared body of the two reference-named classes: same keywords as the reference's constructors; with
    ``csv_file == ""`` the items are synthetic.  ``n_pairs`` (extra keyword, or env VTC_SYNTHETIC_PAIRS) sizes
    the synthetic split; ``resolution`` / ``context`` exist for small test architectures."""
    _kind = "image"
    _default_pairs = 1024

    def _setup(self, csv_file, train, test, add_comments, num_comms, n_pairs, seed, resolution, context, nframes=8):
        if csv_file not in ("", None):
            raise NotImplementedError(
                f"{type(self).__name__}: reading {csv_file!r} needs the reference's dataset_loaders (torchvision video "
                "IO, PIL, CLIP tokenizer, RAKE), which are outside the hot path; an empty csv_file -- what the "
                "reference's configs ship with -- selects synthetic pairs of the same tensor contract")
        if test:
            assert not train                                   # dataset_loaders.py:204-205
        n = int(n_pairs or os.environ.get("VTC_SYNTHETIC_PAIRS", self._default_pairs))
        import warnings
        warnings.warn(f"{type(self).__name__}: csv_file is empty -- {n} SYNTHETIC pairs (random pixels and tokens) of the reference's "
                      "tensor contract stand in for the dataset; retrieval metrics on them say nothing about a real split "
                      "(eval results carry \"synthetic\": true)", stacklevel=3)
        self.synthetic = True
        self.train = train
        self.with_comments = _should_add_comments(add_comments, train)
        nc = int(num_comms) if self.with_comments else 0
        # the three splits are disjoint synthetic sets
        super().__init__(n_pairs=n, kind=self._kind, add_comments=add_comments, num_comms=max(nc, 1),
                         seed=int(seed) + (0 if test else (1 if train else 2)), nframes=nframes, resolution=resolution,
                         context=context)
        if nc == 0:
            # `comments_tok = self._tokenise([""])` (dataset_loaders.py:1024-1025, :562-563): one empty comment
            self.comments = torch.zeros(self.n, 1, context, dtype=torch.int64)
            self.comments[:, 0, 0], self.comments[:, 0, 1] = SOT, EOT


class ImTextDataset(_ReferenceNamedDataset):
    """dataset_loaders/dataset_loaders.py:924-1046 (constructor keywords :938-954)."""
    _kind = "image"

    def __init__(self, csv_file, root="", train=True, test=False, add_comments="train_only", num_comms=0,
                 comment_sampling="random", cached_vision_features=None, test_on_over_k_comms=None, test_set_limit=None,
                 use_augmentation=False, cached_audio_features=None, audio_with_comms=None, audio_instead_of_title=False,
                 n_pairs=None, seed=123, resolution=224, context=77):
        if cached_audio_features is not None or audio_with_comms or audio_instead_of_title:
            raise NotImplementedError("ImTextDataset: the audio branch (needs the external GDT repository) is out of scope")
        if cached_vision_features is not None:
            raise NotImplementedError("ImTextDataset: cached features are produced/consumed by vtc_amd.host.cache_features")
        self.root, self.comment_sampling = root, (comment_sampling if train else None)
        self._setup(csv_file, train, test, add_comments, num_comms, n_pairs, seed, resolution, context)


class VideoDatasetSegments(_ReferenceNamedDataset):
    """dataset_loaders/dataset_loaders.py:440-566 (constructor keywords :449-467): 8-frame segments."""
    _kind = "video"
    _default_pairs = 200

    def __init__(self, csv_file, root="", train=True, test=False, add_comments="train_only", num_comms=2,
                 comment_sampling="random", use_kinetics_train=None, kinetics_csv=None, kinetics_root=None,
                 use_howto100m_train=None, howto100m_csv=None, howto100m_root=None, first_frame_only=False,
                 test_on_over_k_comms=None, test_set_limit=None, n_pairs=None, seed=123, resolution=224, context=77,
                 nframes=8):
        if use_kinetics_train or use_howto100m_train:
            raise NotImplementedError("VideoDatasetSegments: Kinetics / HowTo100M mixing needs real video files")
        self.root, self.first_frame_only = root, first_frame_only
        self._setup(csv_file, train, test, add_comments, num_comms, n_pairs, seed, resolution, context, nframes)

    def __getitem__(self, i):
        vis, title, comments, meta = super().__getitem__(i)
        if self.first_frame_only:                              # dataset_loaders.py:543-544
            vis = vis[0]
        return vis, title, comments, meta


# ---- stand-ins for the video-retrieval benchmark loaders (evaluation/retrieval_evaluation.py:119-134) ----------------------------
#: defaults of the synthetic video stand-ins below; tests on small architectures override them (resolution / context)
VIDEO_STANDIN = {"n_videos": 24, "min_frames": 40, "max_frames": 420, "resolution": 224, "context": 77, "seed": 123}


class _SyntheticVideos(Dataset):
    """Whole videos of ragged length, as the reference's benchmark loaders return them to ``retrieval_evaluation``
    (dataset_loaders/video_retrieval_videodatasets.py:200-256, 528-553; dataset_loaders.py:1080-1111): items are
    ``(frames [T,3,H,W] fp32, captions [1,77] int64, id)`` or, for the sets that carry comments,
    ``(frames, title [1,77], comments [nc,77], id)``.  Those loaders read MSR-VTT / MSVD / Kinetics / Reddit / Livebot video files
    with torchvision + ffmpeg (out of scope); these yield random pixels and tokens of the same tensor contract: T ~ U{min..max}
    frames (so 8-frame chunk counts are ragged at the default stride of 16, with resampled tails), ONE caption per video
    (with several the reference hands faiss a 3-D array).  ``synthetic = True`` marks the results."""
    synthetic = True
    with_comments = False
    n_comments = 5

    def __init__(self, train=False, split="full-test", **kw):
        assert not train, "benchmark stand-ins are evaluation splits"
        cfg = dict(VIDEO_STANDIN)
        cfg.update({k: v for k, v in kw.items() if k in cfg})
        self.cfg, self.split = cfg, split
        self.n = int(os.environ.get("VTC_SYNTHETIC_VIDEOS", cfg["n_videos"]))
        import warnings
        warnings.warn(f"{type(self).__name__}: {self.n} SYNTHETIC videos (random pixels and tokens) of the reference's tensor contract stand "
                      "in for the benchmark; retrieval metrics on them say nothing about the real split", stacklevel=3)
        # a split / class gets its own disjoint synthetic set
        self.seed = int(cfg["seed"]) + sum(ord(c) for c in type(self).__name__ + str(split))
        g = torch.Generator().manual_seed(self.seed)
        self.lengths = torch.randint(cfg["min_frames"], cfg["max_frames"] + 1, (self.n,), generator=g)
        self.captions = synth_tokens(self.n, cfg["context"], g)
        self.comments = synth_tokens(self.n * self.n_comments, cfg["context"], g, empty_frac=0.1).reshape(self.n, self.n_comments, -1)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        r = self.cfg["resolution"]
        frames = torch.randn((int(self.lengths[i]), 3, r, r), generator=g)
        if self.with_comments:
            return frames, self.captions[i][None], self.comments[i], f"video{i}"
        return frames, self.captions[i][None], f"video{i}"


class VideoDatasetMSRVTT(_SyntheticVideos):
    """dataset_loaders/video_retrieval_videodatasets.py:113-256 (non-augmented eval item: ``vid, text, vid_id``)."""


class VideoDatasetMSVD(_SyntheticVideos):
    """dataset_loaders/video_retrieval_videodatasets.py:258-476."""


class VideoDatasetK700Comments(_SyntheticVideos):
    """dataset_loaders/video_retrieval_videodatasets.py:478-556 (``vid, title_tok, comments_tok, vid_id``)."""
    with_comments = True


class VideoDatasetReddit(_SyntheticVideos):
    """dataset_loaders/dataset_loaders.py:1049-1113 (8 frames per item, 5 comments)."""
    with_comments = True

    def __init__(self, train=False, split="test", **kw):
        super().__init__(train=train, split=split, **kw)
        self.lengths = torch.full((self.n,), 8)


class VideoDatasetLivebot(_SyntheticVideos):
    """dataset_loaders/dataset_loaders.py:1116-1180."""
    with_comments = True
