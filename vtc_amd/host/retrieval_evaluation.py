"""The reference's video-benchmark evaluation loop (evaluation/retrieval_evaluation.py:108-268),
served by the HIP path -- with the per-video batch-1 loop replaced by ragged batches.

Reference semantics kept (file:line):
  * frames[:, ::frame_stride] -> 8-frame chunks, a short tail resampled by floor(linspace)   (:174-199)
  * CAM models get comments: dummy ``tokenize([""] * 5)`` rows ([SOT, EOT, 0...], replaced by the
    model's mask_embedding) or the first 5 real comments, repeated per adapted item           (:203-231)
  * video embedding = mean over its chunk embeddings, NOT re-normalised                       (:254-259)
  * compute_recall: RecallAtK([1,5,10]) both directions, x100                                 (:23-47)
What changes: instead of one ``model.forward`` per video (DataLoader batch_size=1, :136), the chunks of
many videos are encoded in one tower call, the captions in another, and a segment-mean kernel
(vtc_segment_mean) reduces the ragged chunk groups.  The towers are per-item functions, so the
embeddings are the same as the per-video loop's.  Only the one-caption-per-video case is defined
(SURVEY 3.3 caveat: the reference hands a 3-D array to faiss otherwise).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops
from . import model as module_arch
from .metric import RecallAtK

SOT, EOT = 49406, 49407
models_needing_comments = (module_arch.PretrainedCLIP_finaltf, module_arch.PretrainedCLIP_TimeSformer_finaltf)


def empty_comments(n: int, n_comments: int = 5, ctx: int = 77) -> torch.Tensor:
    """clip.tokenize([""] * n_comments) stacked n times (:214): [SOT, EOT, 0...] rows."""
    c = torch.zeros(n, n_comments, ctx, dtype=torch.int64)
    c[..., 0] = SOT
    c[..., 1] = EOT
    return c


def chunk_frames(frames: torch.Tensor, frame_stride: int = 16, nframes: int = 8, first_chunk_only: bool = False) -> torch.Tensor:
    """[T,3,H,W] -> [nchunks, nframes, 3, H, W]   (:179-196)."""
    frames = frames[::frame_stride]
    out = []
    for x in torch.split(frames, nframes, 0):
        if x.shape[0] != nframes:
            idx = torch.floor(torch.linspace(0, x.shape[0] - 1, nframes)).to(torch.int64).to(x.device)
            x = torch.index_select(x, 0, idx)
        out.append(x)
    chunks = torch.stack(out)
    return chunks[0:1] if first_chunk_only else chunks


def compute_recall(video_emb, caption_emb, precision=None):
    """compute_recall (:23-47): returns {"Video to Text": [R@1,R@5,R@10], "Text to Video": [...]} in percent,
    named exactly as the reference's DataFrame columns (tvr / vtr, :36-43)."""
    m = RecallAtK("videos", "titles", [1, 5, 10])
    if precision is not None:
        m.precision = precision
    r_v2t, r_t2v = m.compute_both(video_emb, caption_emb)        # compute(video, caption), compute(caption, video)
    vtr = np.array(r_v2t)[:, 1] * 100.0
    tvr = np.array(r_t2v)[:, 1] * 100.0
    return {"Video to Text": tvr, "Text to Video": vtr, "index": ["R@1", "R@5", "R@10"]}


@torch.no_grad()
def encode_videos(model, videos: Sequence[Tuple], device, frame_stride: int = 16, first_chunk_only: bool = False,
                  max_chunks_per_call: int = 256):
    """videos: sequence of (frames [T,3,H,W], caption [77] int64[, comments [nc,77] int64]).
    Returns (video_emb [N,D] = mean of chunk embeddings, caption_emb [N,D]) on the GPU."""
    model.eval()
    cam = isinstance(model, models_needing_comments)
    chunks, captions, comments, counts = [], [], [], []
    for item in videos:
        fr, cap = item[0], item[1]
        assert fr.dim() == 4 and fr.shape[1] == 3 and cap.dim() == 1, "one caption per video, frames [T,3,H,W]"
        ch = chunk_frames(fr, frame_stride, 8, first_chunk_only)
        chunks.append(ch)
        counts.append(ch.shape[0])
        captions.append(cap)
        if cam:
            comments.append(item[2][:5] if len(item) > 2 and item[2] is not None else empty_comments(1, 5, cap.shape[0])[0])
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=device)
    caps = torch.stack(captions).to(device)
    all_chunks = torch.cat(chunks)
    fv = torch.cat([model.encode_image(all_chunks[i:i + max_chunks_per_call].to(device))
                    for i in range(0, all_chunks.shape[0], max_chunks_per_call)])
    ft = model.encode_text(caps)
    if cam and model.branch_to_adapt_val != "skip":
        comm = torch.stack(comments).to(device)                                  # [N, nc, ctx]
        packed = model._pack()["cam"]
        if model.branch_to_adapt_val == "text":                                   # one comment set per caption (:207-210)
            fc = model.encode_text(comm.reshape(-1, comm.shape[-1]))
            ft = packed.forward(ft, fc, comm)
        else:                                                                     # "image": per chunk (:207-208)
            rep = torch.repeat_interleave(torch.arange(len(counts), device=device), torch.tensor(counts, device=device))
            comm_c = comm[rep]
            fc = model.encode_text(comm.reshape(-1, comm.shape[-1])).reshape(len(counts), comm.shape[1], -1)[rep]
            fv = packed.forward(fv, fc.reshape(-1, fc.shape[-1]).contiguous(), comm_c.contiguous())
    fv, ft = ops.normalize_rows(fv), ops.normalize_rows(ft)                        # forward()'s final normalize
    return ops.segment_mean(fv, offsets), ft


@torch.no_grad()
def retrieval_evaluation(model, videos: Sequence[Tuple], device="cuda", frame_stride: int = 16, first_chunk_only: bool = False):
    video_emb, caption_emb = encode_videos(model, videos, device, frame_stride, first_chunk_only)
    return compute_recall(video_emb, caption_emb), video_emb, caption_emb
