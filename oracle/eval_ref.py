"""CPU restatement of model/metric.py RecallAtK.compute and the eval-harness bookkeeping.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``RecallAtK.compute`` delegates the search to ``faiss.GpuIndexFlatL2``
(faiss-gpu, environment.yml:12, un-pinned, not installed here and GPU-only), so the
reference's class cannot run in this container; nothing in the reference's tests
pins it either: PARITY UNPINNED by the reference.  This file follows
model/metric.py:137-161 literally; the search itself is restated from faiss's
published exact-L2 algorithm (squared L2 = |q|^2 + |g|^2 - 2 q.g in fp32, k
smallest).  Tie order is unspecified in faiss; this build defines it as lowest
gallery index first, and ``near_ties`` reports when a case depends on it.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch


def l2_topk(features_a: np.ndarray, features_b: np.ndarray, depth: int, dtype=np.float32,
            row_block: int = 2048) -> Tuple[np.ndarray, np.ndarray]:
    """index.add(features_a); index.search(features_b, depth) (model/metric.py:140-146).

    Returns (ids [Nb, depth] int64, dists [Nb, depth]) of the ``depth`` nearest rows of
    ``features_a`` for every row of ``features_b`` by squared L2, ascending, ties by lowest
    index.  ``dtype=np.float64`` gives ground-truth ranks."""
    a = np.ascontiguousarray(features_a, dtype=dtype)
    b = np.ascontiguousarray(features_b, dtype=dtype)
    an = (a * a).sum(1)
    ids = np.empty((b.shape[0], depth), dtype=np.int64)
    ds = np.empty((b.shape[0], depth), dtype=dtype)
    for s in range(0, b.shape[0], row_block):
        q = b[s:s + row_block]
        d = (q * q).sum(1)[:, None] + an[None, :] - dtype(2) * (q @ a.T)
        # stable sort => equal distances keep ascending index order
        order = np.argsort(d, axis=1, kind="stable")[:, :depth]
        ids[s:s + row_block] = order
        ds[s:s + row_block] = np.take_along_axis(d, order, axis=1)
    return ids, ds


def recall_from_ids(ids: np.ndarray, k_vals: Sequence[int], num_samples: int) -> List[Tuple[int, float]]:
    """model/metric.py:148-160: hit if ``target in rp[:k]`` where target is the query's own
    row index; denominator is len(features_a) (:138,:158)."""
    out = []
    tgt = np.arange(ids.shape[0])[:, None]
    for k in k_vals:
        hits = int((ids[:, :k] == tgt).any(axis=1).sum())
        out.append((k, hits / num_samples))
    return out


def recall_at_k(features_a: np.ndarray, features_b: np.ndarray, k_vals: Sequence[int] = (1, 5, 10),
                dtype=np.float32) -> List[Tuple[int, float]]:
    """RecallAtK(k_vals).compute(features_a, features_b), model/metric.py:137-161.
    Search depth is max(k)+1 (:145)."""
    depth = int(np.max(k_vals) + 1)
    ids, _ = l2_topk(features_a, features_b, min(depth, features_a.shape[0]), dtype)
    return recall_from_ids(ids, k_vals, features_a.shape[0])


def near_ties(features_a: np.ndarray, features_b: np.ndarray, k_vals: Sequence[int] = (1, 5, 10),
              tol: float = 1e-6) -> int:
    """Number of queries whose R@K outcome depends on a distance gap < tol (fp64):
    the target sits at rank k or k+1 and its neighbour across the boundary is closer than tol."""
    depth = int(np.max(k_vals) + 1)
    depth = min(depth + 1, features_a.shape[0])
    ids, ds = l2_topk(features_a, features_b, depth, np.float64)
    n = 0
    for i in range(ids.shape[0]):
        pos = np.nonzero(ids[i] == i)[0]
        if len(pos) == 0:
            continue
        r = int(pos[0])
        for k in k_vals:
            if r == k - 1 and k < depth and abs(ds[i, k] - ds[i, r]) < tol:
                n += 1
            elif r == k and abs(ds[i, r] - ds[i, k - 1]) < tol:
                n += 1
    return n


def eval_result_dict(res_vis: np.ndarray, res_text: np.ndarray, dtype=np.float32) -> dict:
    """evaluation/eval.py:121-138: both directions, k in {1,5,10}, JSON keys.  ``dtype=np.float64``: ground-truth ranks."""
    t_from_i = recall_at_k(res_vis, res_text, [1, 5, 10], dtype)
    i_from_t = recall_at_k(res_text, res_vis, [1, 5, 10], dtype)
    return {
        "R1_title_from_im": t_from_i[0][1], "R5_title_from_im": t_from_i[1][1], "R10_title_from_im": t_from_i[2][1],
        "R1_im_from_title": i_from_t[0][1], "R5_im_from_title": i_from_t[1][1], "R10_im_from_title": i_from_t[2][1],
    }


def compute_recall_table(tensor_v: torch.Tensor, tensor_t: torch.Tensor, dtype=np.float32):
    """compute_recall, evaluation/retrieval_evaluation.py:23-47 (one caption per video:
    ``tensor_t.numpy().squeeze()`` must be 2-D, SURVEY 3.3 caveat).  Returns
    (video_to_text[3], text_to_video[3]) in percent for R@1/5/10, named as the reference's
    DataFrame columns name them (:39-43: 'Video to Text' = tvr, 'Text to Video' = vtr)."""
    t = tensor_t.numpy().squeeze()
    assert t.ndim == 2, "only the one-caption-per-video case is defined"
    vtr = np.array(recall_at_k(tensor_v.numpy(), t, [1, 5, 10], dtype))[:, 1] * 100.0
    tvr = np.array(recall_at_k(t, tensor_v.numpy(), [1, 5, 10], dtype))[:, 1] * 100.0
    return tvr, vtr


def chunk_frames(frames: torch.Tensor, frame_stride: int = 16, nframes: int = 8) -> torch.Tensor:
    """evaluation/retrieval_evaluation.py:174-199: [1,T,3,H,W] -> [nchunks, 8, 3, H, W];
    stride, split into 8-frame chunks, a short tail is resampled by floor(linspace)."""
    frames = frames[:, ::frame_stride]
    out = []
    for x in torch.split(frames, nframes, 1):
        if x.shape[1] != nframes:
            idx = torch.floor(torch.linspace(0, x.shape[1] - 1, nframes)).to(torch.int64)
            x = torch.index_select(x, 1, idx)
        out.append(x)
    return torch.cat(out, dim=0)


def mean_chunks(video_embeddings: Sequence[torch.Tensor]) -> torch.Tensor:
    """evaluation/retrieval_evaluation.py:254-259: per-video mean over chunk embeddings,
    NOT re-normalised."""
    return torch.cat([torch.mean(k, dim=0, keepdim=True) for k in video_embeddings])


def retrieval_evaluation_loop(forward, items, needs_comments: bool, branch: str = "text", frame_stride: int = 16,
                              first_frame_only: bool = False, first_chunk_only: bool = False, n_comments: int = 5):
    """The reference's per-video batch-1 loop, evaluation/retrieval_evaluation.py:143-264, line by line.

    ``forward(frames, captions, comments | None) -> (feats_a, feats_b)`` is the model's forward (an oracle wrapper of model_ref.py);
    ``items`` are dataset items ``(frames [T,3,H,W], captions [ncap,77], id)`` or ``(frames, captions, comments [nc,77], id)``; the
    DataLoader's batch dimension of one (:136) is added here.  Returns (video_joint_tensor [N,D], caption_joint_tensor [N,ncap,D])
    as :254-260 build them: the mean over a video's chunk embeddings, NOT re-normalised."""
    video_joint_embeddings, caption_joint_embeddings = [], []
    for it in items:
        if len(it) == 3:                                                   # :144-150
            frames, captions, comments = it[0][None], it[1][None], None
        else:
            frames, captions, comments = it[0][None], it[1][None], it[2][None]
        assert captions.dim() == 3 and captions.shape[0] == 1              # :159
        assert frames.dim() == 5 and frames.shape[0] == 1 and frames.shape[2] == 3
        captions = captions[0]                                             # :163
        if first_frame_only:                                               # :165-173 (the isinstance test of :166 is never true: every
            frames = frames[0][0:1]                                        #  image model is also listed in video_models, :56-62)
            assert not first_chunk_only
        else:                                                              # :175-199
            chunks = chunk_frames(frames, frame_stride, 8)
            if first_chunk_only:
                chunks = chunks[0:1]
            frames = chunks
        if needs_comments:                                                 # :203-231
            ncomms = len(frames) if branch == "image" else len(captions)
            if comments is None:
                dummy = torch.zeros(n_comments, captions.shape[-1], dtype=torch.int64)      # clip.tokenize([""] * 5): [SOT, EOT, 0...]
                dummy[:, 0], dummy[:, 1] = 49406, 49407
                comments = torch.stack([dummy for _ in range(ncomms)])
            else:
                comments = comments[0, :n_comments]
                comments = torch.stack([comments for _ in range(ncomms)])
            feats_a, feats_b = forward(frames, captions, comments)
        else:
            feats_a, feats_b = forward(frames, captions, None)
        video_joint_embeddings.append(feats_a)
        caption_joint_embeddings.append(feats_b)
    video_joint_tensor = mean_chunks(video_joint_embeddings)              # :254-259
    caption_joint_tensor = torch.stack(caption_joint_embeddings)           # :260 (equal caption counts)
    return video_joint_tensor, caption_joint_tensor


def compute_recall_frame(tensor_v: torch.Tensor, tensor_t: torch.Tensor, split: str = "full-test", dataset_name: str = "MSRVTT",
                         dtype=np.float32):
    """compute_recall's DataFrame (:23-47), columns and index named as the reference names them."""
    import pandas as pd
    tvr, vtr = compute_recall_table(tensor_v, tensor_t, dtype)
    return pd.DataFrame({f"{dataset_name} {split} split Video to Text": tvr, f"{dataset_name} {split} split Text to Video": vtr},
                        index=[f"R@{i}" for i in (1, 5, 10)])
