"""Drop-in for the reference's ``evaluation/retrieval_evaluation.py`` -- the video-benchmark evaluation
the trainer runs at every validation epoch (trainer/trainer.py:159-173) -- served by the HIP path.

Same module surface as the reference (file:line):

    compute_recall(tensor_v, tensor_t, split, dataset_name) -> pandas.DataFrame      (:23-47)
    models_needing_comments, image_models, video_models                               (:50-62)
    load_model(checkpoint_path, device, model_type)                                   (:65-105)
    retrieval_evaluation(model, datasetname, split, device, out_csv=None,
                         frame_stride=16, first_frame_only=False, first_chunk_only=False)   (:108-268)
    the ``python evaluation/retrieval_evaluation.py -c ... -m ...`` flags             (:271-360)

Reference semantics kept:
  * frames[:, ::frame_stride] -> 8-frame chunks, a short tail resampled by floor(linspace)   (:174-199)
  * ``first_frame_only``: the first frame alone, as a 4-D image batch of one                   (:165-173)
  * image wrappers (PretrainedCLIP, PretrainedCLIP_finaltf) are ``video_models`` too (:56-62), so without
    ``first_frame_only`` they receive the 5-D chunks and average the per-frame ViT features of a chunk
    (model/model.py:333-338,465-470)
  * CAM models get comments: dummy ``tokenize([""] * 5)`` rows ([SOT, EOT, 0...], replaced by the
    model's mask_embedding) or the first 5 real comments, repeated per adapted item           (:203-231)
  * video embedding = mean over its chunk embeddings, NOT re-normalised                       (:254-259)
  * compute_recall: RecallAtK([1,5,10]) both directions, x100, the reference's column names   (:23-47)
What changes, all inside the hot path's boundary: instead of one ``model.forward`` per video (DataLoader
batch_size=1, :136), the chunks of many videos go through one tower call, the captions through another,
and a segment-mean kernel (vtc_segment_mean) reduces the ragged chunk groups; embeddings stay on the GPU.
The towers and the CAM are per-item functions, so the embeddings equal the per-video loop's
(tests/test_gpu_retrieval_eval.py holds them to the oracle's batch-1 loop).  The dataset names resolve to
synthetic stand-ins of the same tensor contract (vtc_amd/host/datasets.py), as ``eval.py`` does for
``ImTextDataset``.  Only the one-caption-per-video case is defined (with several captions the
reference hands a 3-D array to faiss).
"""
from __future__ import annotations

import argparse
import logging
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops
from . import datasets as module_data
from . import model as module_arch
from .metric import RecallAtK

SOT, EOT = 49406, 49407

models_needing_comments = (module_arch.PretrainedCLIP_finaltf, module_arch.PretrainedCLIP_TimeSformer_finaltf)
image_models = (module_arch.PretrainedCLIP, module_arch.PretrainedCLIP_finaltf)
video_models = (module_arch.PretrainedCLIP, module_arch.PretrainedCLIP_finaltf, module_arch.PretrainedCLIP_TimeSformer,
                module_arch.PretrainedCLIP_TimeSformer_finaltf)

#: the CLI's parsed flags (the reference's ``load_model`` reads a module-level ``args``, :76,:87)
args = argparse.Namespace(branch_to_adapt="text", residual_activation="none")

_DATASETS = {"MSRVTT_videos": "VideoDatasetMSRVTT", "MSVD_videos": "VideoDatasetMSVD", "K700_videos": "VideoDatasetK700Comments",
             "Reddit_videos": "VideoDatasetReddit", "livebot": "VideoDatasetLivebot"}


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


def compute_recall(tensor_v, tensor_t, split: str = "full-test", dataset_name: str = "MSRVTT", precision=None):
    """compute_recall (:23-47): a DataFrame with the rows R@1 / R@5 / R@10 (percent) and the reference's two
    columns ``f"{dataset_name} {split} split Video to Text"`` / ``"... Text to Video"`` (tvr / vtr, :36-43).
    Takes the reference's CPU tensors (``[N, D]`` and ``[N, 1, D]``) as well as GPU ones."""
    import pandas as pd
    tv, tt = torch.as_tensor(tensor_v), torch.as_tensor(tensor_t)
    if tt.dim() == 3 and tt.shape[1] == 1:                       # tensor_t.numpy().squeeze() of one caption per video
        tt = tt[:, 0]
    if tv.dim() != 2 or tt.dim() != 2:
        raise ValueError("compute_recall: one caption per video only (the reference hands faiss a 3-D array otherwise); "
                         f"got video {tuple(tv.shape)}, captions {tuple(tt.shape)}")
    recall_range = [1, 5, 10]
    m = RecallAtK("videos", "titles", recall_range)
    if precision is not None:
        m.precision = precision
    r_v2t, r_t2v = m.compute_both(tv, tt)                        # compute(video, caption), compute(caption, video)
    vtr = np.array(r_v2t)[:, 1] * 100.0
    tvr = np.array(r_t2v)[:, 1] * 100.0
    df = pd.DataFrame({f"{dataset_name} {split} split Video to Text": tvr,
                       f"{dataset_name} {split} split Text to Video": vtr}, index=[f"R@{i}" for i in recall_range])
    logging.info(df)
    return df


def load_model(checkpoint_path: Optional[str], device: str, model_type: str, branch_to_adapt: Optional[str] = None,
               residual_activation: Optional[str] = None):
    """load_model (:65-105).  ``branch_to_adapt`` / ``residual_activation`` default to the CLI's parsed flags (the
    reference reads them from its module-level ``args``)."""
    branch_to_adapt = args.branch_to_adapt if branch_to_adapt is None else branch_to_adapt
    residual_activation = args.residual_activation if residual_activation is None else residual_activation
    init_from_avg = False
    checkpoint = None
    if checkpoint_path is not None:
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
        init_from_avg = checkpoint["config"]["arch"]["args"].get("init_from_avg", False)
    if model_type == "pretrained_clip":
        model = module_arch.PretrainedCLIP(model_type="ViT-B/32", freeze=False, residual_activation=residual_activation)
    elif model_type == "clip_timesformer":
        model = module_arch.PretrainedCLIP_TimeSformer(residual_activation=residual_activation)
    elif model_type == "pretrained_clip_finaltf":
        model = module_arch.PretrainedCLIP_finaltf(branch_to_adapt_val=branch_to_adapt, residual_activation=residual_activation,
                                                   init_from_avg=init_from_avg)
    elif model_type == "clip_timesformer_finaltf":
        model = module_arch.PretrainedCLIP_TimeSformer_finaltf(branch_to_adapt_val=branch_to_adapt,
                                                               residual_activation=residual_activation, init_from_avg=init_from_avg)
    else:
        raise Exception(f"Unknown model_type {model_type!r}")     # (the reference falls through to an unbound `model`)
    if checkpoint is not None:
        model.load_state_dict(checkpoint["state_dict"])
    model.eval()
    model.to(device)
    return model


def _item_parts(item):
    """A dataset item as the reference's loaders return it: (frames, captions, id) or (frames, captions, comments, id)
    (:144-150; the DataLoader's batch dimension of one is not added here).  captions: [77] or [1, 77]."""
    if len(item) == 3 and not torch.is_tensor(item[2]) and item[2] is not None:
        fr, cap, com = item[0], item[1], None
    elif len(item) == 3:                                          # (frames, caption, comments | None): the list form of the tests
        fr, cap, com = item
    elif len(item) == 4:
        fr, cap, com = item[0], item[1], item[2]
    else:
        fr, cap, com = item[0], item[1], None
    if cap.dim() == 2:
        assert cap.shape[0] == 1, "one caption per video (SURVEY 3.3: several captions reach faiss as a 3-D array in the reference)"
        cap = cap[0]
    assert fr.dim() == 4 and fr.shape[1] == 3 and cap.dim() == 1, "frames [T,3,H,W], caption [77]"      # :159-160
    return fr, cap, com


@torch.no_grad()
def encode_videos(model, videos: Sequence[Tuple], device, frame_stride: int = 16, first_chunk_only: bool = False,
                  max_chunks_per_call: int = 256, first_frame_only: bool = False):
    """videos: sequence of dataset items (see _item_parts).  Returns (video_emb [N,D] = mean of the video's chunk
    embeddings, caption_emb [N,D]) on the GPU -- the two tensors the reference stacks at :254-260."""
    model.eval()
    if not isinstance(model, video_models):
        raise Exception("Unknown model_type")                      # :200-201
    cam = isinstance(model, models_needing_comments)
    timesformer = getattr(model, "_video_tower", False)
    if first_frame_only:
        assert not first_chunk_only                                  # :173
        if timesformer:
            raise ValueError("first_frame_only hands a 4-D [1,3,H,W] batch to forward(); the TimeSformer wrappers take "
                             "[B,F,3,H,W] only (model/timesformer_clip_alt.py:253 unpacks five dimensions)")
    chunks, captions, comments, counts = [], [], [], []
    for item in videos:
        fr, cap, com = _item_parts(item)
        ch = fr[0:1] if first_frame_only else chunk_frames(fr, frame_stride, 8, first_chunk_only)   # [1,3,H,W] | [C,8,3,H,W]
        chunks.append(ch)
        counts.append(ch.shape[0])
        captions.append(cap)
        if cam:
            comments.append(com[:5] if com is not None else empty_comments(1, 5, cap.shape[0])[0])
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=device)
    caps = torch.stack(captions).to(device)
    all_chunks = torch.cat(chunks)
    tower = model._pack()["visual"]
    if timesformer or first_frame_only:
        # video tower on [C,8,3,H,W], or the image tower on single frames
        fv = torch.cat([tower.forward(all_chunks[i:i + max_chunks_per_call].to(device))
                        for i in range(0, all_chunks.shape[0], max_chunks_per_call)])
    else:
        # image wrapper on 5-D chunks: per-frame ViT, mean over the chunk's frames (model/model.py:333-338,465-470)
        nfr = all_chunks.shape[1]
        step = max(1, max_chunks_per_call * 8 // nfr)
        fv = torch.cat([ops.mean_groups(tower.forward(all_chunks[i:i + step].to(device).flatten(0, 1)), nfr)
                        for i in range(0, all_chunks.shape[0], step)])
    ft = model.encode_text(caps)
    if cam and model.branch_to_adapt_val != "skip":
        comm = torch.stack(comments).to(device)                                  # [N, nc, ctx]
        packed = model._pack()["cam"]
        if model.branch_to_adapt_val == "text":                                   # one comment set per caption (:207-210)
            fc = model.encode_text(comm.reshape(-1, comm.shape[-1]))
            ft = packed.forward(ft, fc, comm)
        elif model.branch_to_adapt_val == "image":                                # per chunk (:207-208)
            rep = torch.repeat_interleave(torch.arange(len(counts), device=device), torch.tensor(counts, device=device))
            comm_c = comm[rep]
            fc = model.encode_text(comm.reshape(-1, comm.shape[-1])).reshape(len(counts), comm.shape[1], -1)[rep]
            fv = packed.forward(fv, fc.reshape(-1, fc.shape[-1]).contiguous(), comm_c.contiguous())
        else:
            raise Exception("Unknown branch_to_adapt")                           # model/model.py:261
    fv, ft = ops.normalize_rows(fv), ops.normalize_rows(ft)                        # forward()'s final normalize
    v_emb = ops.segment_mean(fv, offsets)
    module_arch.raise_if_nonfinite("retrieval_evaluation", v_emb, ft)
    return v_emb, ft


def _resolve_dataset(datasetname, split):
    if not isinstance(datasetname, str):
        return datasetname, "videos"                                 # a Dataset / sequence of items (extension)
    if datasetname not in _DATASETS:
        raise Exception("Unknown dataset")                           # :133-134
    return getattr(module_data, _DATASETS[datasetname])(train=False, split=split), datasetname


@torch.no_grad()
def retrieval_evaluation(model, datasetname, split: str = "full-test", device="cuda", out_csv: Optional[str] = None,
                         frame_stride: int = 16, first_frame_only: bool = False, first_chunk_only: bool = False,
                         videos_per_batch: int = 64, return_embeddings: bool = False):
    """retrieval_evaluation (:108-268) with the reference's positional arguments; returns its DataFrame.
    ``datasetname`` may also be a Dataset or a list of items (frames, captions[, comments], id).
    ``videos_per_batch`` bounds how many videos are decoded before their chunks are encoded (host memory);
    ``return_embeddings``: also the two stacked embedding tensors of :254-260 (GPU)."""
    dataset, name = _resolve_dataset(datasetname, split)
    v_parts, c_parts = [], []
    batch = []
    n = len(dataset)
    for i in range(n):
        batch.append(dataset[i])
        if len(batch) == videos_per_batch or i == n - 1:
            v, c = encode_videos(model, batch, device, frame_stride, first_chunk_only, first_frame_only=first_frame_only)
            v_parts.append(v)
            c_parts.append(c)
            batch = []
    video_emb, caption_emb = torch.cat(v_parts), torch.cat(c_parts)
    outdf = compute_recall(video_emb, caption_emb, split=split, dataset_name=name)
    if getattr(dataset, "synthetic", False):
        outdf.attrs["synthetic"] = True
    if out_csv is not None:
        outdf.to_csv(out_csv)
    return (outdf, video_emb, caption_emb) if return_embeddings else outdf


def cli(argv=None):
    """The reference's flags (:271-360)."""
    global args
    ap = argparse.ArgumentParser(description="VTC video retrieval evaluation (MI355X)")
    ap.add_argument("-c", "--dataset", default="MSRVTT_videos", choices=list(_DATASETS), type=str, help="dataset to load")
    ap.add_argument("-r", "--checkpoint", default=None, type=str, help="path to checkpoint (default: None)")
    ap.add_argument("-m", "--model_type", default=None, type=str, help="model arch to be loaded")
    ap.add_argument("-d", "--device", default="cuda", type=str, help="device to load model on")
    ap.add_argument("-s", "--split", default="full-test", type=str, help="which test split to use")
    ap.add_argument("--branch_to_adapt", default="text", choices=["text", "image", "random", "skip"], type=str,
                    help="which branch to adapt for finaltf models")
    ap.add_argument("--residual_activation", default="none", type=str, help="which activation fn to use on the residual")
    ap.add_argument("--out_csv", default=None, type=str, help="File to save output csv")
    ap.add_argument("--frame_stride", default=16, type=int, help="Video frame stride")
    ap.add_argument("--first_frame_only", action="store_true", help="Use only the first frame of a video, as if it were an image")
    ap.add_argument("--first_chunk_only", action="store_true", help="Use only the first 8-frame chunk of a video")
    args = ap.parse_args(argv)
    model = load_model(args.checkpoint, args.device, model_type=args.model_type)
    df = retrieval_evaluation(model, args.dataset, args.split, args.device, out_csv=args.out_csv, frame_stride=args.frame_stride,
                              first_frame_only=args.first_frame_only, first_chunk_only=args.first_chunk_only)
    print(df)
    return df
