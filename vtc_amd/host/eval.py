"""Drop-in for the reference's ``evaluation/eval.py`` entry point (evaluation/eval.py:50-196):

    python evaluation/eval.py -c configs/pretrained_clip_comments_attention.jsonc [-r ckpt] [-d 0]
                              [--bs N] [--bv branch] [--nc n] [--am fusion] [--ac mode] [--dtype bf16|f32]

Same flags, same result JSON keys (R{1,5,10}_title_from_im / _im_from_title, :131-138), same
loop semantics (encode every pair, stack, Recall@K both directions).  Differences, all inside the
hot path's boundary: embeddings stay on the GPU between batches instead of a D2H per batch
(:114-115), the k-NN runs in libvtc_hip.so instead of faiss, and the dataset ``type`` may be
``SyntheticPairs``; the reference's own ``ImTextDataset`` / ``VideoDatasetSegments`` names resolve too and, with the
empty ``csv_file`` the reference's configs ship with, yield synthetic pairs of the same tensor contract
(vtc_amd/host/datasets.py), so ``configs/pretrained_clip*.jsonc`` run unmodified."""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys

import torch
from torch.utils.data import DataLoader

from . import datasets as module_data
from . import model as module_arch
from .metric import RecallAtK
from .parse_config import ConfigParser


def add_irrelevant_comms(comments: torch.Tensor, num_irrelevant_comments: int) -> torch.Tensor:
    """evaluation/eval.py:23-47: append ``num_irrelevant_comments`` comments drawn from other items of the batch
    ("all additional comments come from different elements in the batch").  Same numpy draws in the same order
    (comment indices first, then one batch index per comment, re-drawn once when it hits the item itself).  The
    reference's body cannot run as written -- ``torch.cat([comments[i], new_comm_list], 0)`` concatenates a tensor
    with a list and ``return`` sits inside the loop, after item 0 -- so this is its documented intent applied to
    every item, on the token ids (before the hot path)."""
    import numpy as np
    bs, nc, ntok = comments.shape
    out = torch.zeros((bs, nc + num_irrelevant_comments, ntok), dtype=comments.dtype)
    for i in range(bs):
        comm_indices = np.random.randint(low=0, high=nc, size=num_irrelevant_comments)
        extra = []
        for comm_ind in comm_indices:
            batch_ind = int(np.random.randint(low=0, high=bs, size=[1])[0])
            if batch_ind == i:
                batch_ind = int(np.random.randint(low=0, high=bs, size=[1])[0])
            extra.append(comments[batch_ind, comm_ind])
        out[i] = torch.cat([comments[i], torch.stack(extra)], 0) if extra else comments[i]
    return out.long()


def main(config: ConfigParser, args, checkpoint_path=None, device="cuda"):
    """One process: the reference's loop (evaluation/eval.py:50-141).  Under ``python -m torch.distributed.run --nproc-per-node G
    evaluation/eval.py ...`` (WORLD_SIZE > 1; BASELINE configs[3]): rank r encodes the contiguous shard [lo_r, hi_r) of the dataset on its
    own GPU -- no collective in the encode path -- the embeddings are exchanged once and the sweep runs sharded
    (vtc_amd/dist.py sharded_recall: all-gather, one [N/G, N] distance GEMM per rank, all-to-all of column planes, all-reduce of the six
    counters); rank 0 writes the reference's JSON.  The towers and the CAM are per-item functions and the sharded sweep returns the
    single-GPU counters, so the JSON equals the one-process run's (tests/test_gpu_eval_entry.py, world 2 and 3 on one card)."""
    rank, world = 0, int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        from .. import dist as vdist
        rank, local, world = vdist.init_from_env()
        device = f"cuda:{local}"
    dataset = config.init_obj("dataset", module_data, train=False, test=True)
    arch_args = config["arch"].get("args", {})
    branch_to_adapt = arch_args.get("branch_to_adapt_val", None)
    comment_fusion = arch_args.get("comment_fusion", None)
    num_comms = config["dataset"]["args"].get("num_comms", None)
    add_comments = config["dataset"]["args"].get("add_comments", "never")
    if branch_to_adapt is None:
        exp_combo = "title_only" if add_comments != "always" else f"{comment_fusion}_{num_comms}_comms"
    else:
        exp_combo = f"adapted_{branch_to_adapt}_{num_comms}_comms"
    if checkpoint_path is not None:
        save_path = f"{checkpoint_path.absolute().as_posix()[:-4]}_res_{exp_combo}.json"
    else:
        save_path = getattr(args, "out", None) or f"zero_shot_res_{comment_fusion}.json"
    logging.info(f"Saving results to {save_path}")

    n_total = len(dataset)
    shard = dataset
    if world > 1:
        lo, hi = vdist.shard_bounds(n_total, rank, world)
        shard = torch.utils.data.Subset(dataset, range(lo, hi))
        logging.info(f"rank {rank}/{world}: pairs [{lo}, {hi}) of {n_total} on {device}")
    loader = DataLoader(shard, batch_size=config["batch_size"], num_workers=getattr(args, "workers", 0), shuffle=False)
    seed = os.environ.get("VTC_EVAL_SEED", "1023" if world > 1 else None)
    if seed is not None and checkpoint_path is None:
        # randomly initialised towers (no checkpoint, no VTC_CLIP_WEIGHTS) must be the SAME towers on every rank: the reference's seed
        # (train.py:34-38), set right before construction
        torch.manual_seed(int(seed))
    model = config.init_obj("arch", module_arch)
    if checkpoint_path is not None:
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
        model.load_state_dict(checkpoint["state_dict"])            # strict, eval.py:90-91
    if checkpoint_path is None and not module_arch.clip_arch.pretrained_weights_available():
        logging.warning("zero-shot eval WITHOUT pretrained CLIP weights: VTC_CLIP_WEIGHTS is unset, the towers are randomly "
                        "initialised and the recall numbers are meaningless (the reference's clip.load downloads ViT-B/32)")
    dt = module_arch.parse_compute_dtype(getattr(args, "dtype", None))
    if dt is not None:
        model.compute_dtype = dt
    logging.info(f"operand arithmetic: {model.compute_dtype} (the reference computes in fp32, model/model.py:318; "
                 "--dtype f32 or VTC_COMPUTE_DTYPE=f32 reproduces it to 1e-5, the 16-bit default to 1e-3)")
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is not None:
        torch.cuda.set_device(dev)            # launches go to the tensors' device anyway (ops.on_device); this keeps torch's
    model = model.eval().to(device)           # own allocations and the default stream on it too
    num_irrelevant_comments = getattr(args, "num_irrelevant_comments", 0)

    res_vis, res_text = [], []
    with torch.no_grad():
        for vis, title, comments, meta in loader:
            if num_irrelevant_comments:
                assert num_irrelevant_comments <= config["batch_size"], \
                    "Number of irrelevant comments needs to be smaller than batch size."        # eval.py:105-107
                comments = add_irrelevant_comms(comments, num_irrelevant_comments)
            out = model.forward(vis.to(device), title.to(device), comments.to(device))
            res_vis.append(out[0])
            res_text.append(out[1])
    res_vis, res_text = torch.cat(res_vis), torch.cat(res_text)
    # unconditional: an IEEE-half overflow of the text blocks in a batch after the first, or a one-launch CAM that gave up at a
    # grid barrier, returns NaN rows with rc 0 -- such a run raises here instead of writing a JSON
    if world > 1:
        # (the finite check is inside: the flag word rides behind the counters through their all-reduce, so EVERY rank raises)
        phases = {}
        r_ab, r_ba = vdist.sharded_recall(res_vis, res_text, n_total, [1, 5, 10], rank, world, phases=phases)
        t_from_i, i_from_t = [(k, r_ab[k]) for k in (1, 5, 10)], [(k, r_ba[k]) for k in (1, 5, 10)]
        logging.info(f"rank {rank}: sharded sweep phases {phases}")
    else:
        module_arch.raise_if_nonfinite("eval", res_vis, res_text)
        t_from_i, i_from_t = RecallAtK("images", "titles", [1, 5, 10]).compute_both(res_vis, res_text)
    out = {"R1_title_from_im": t_from_i[0][1], "R5_title_from_im": t_from_i[1][1], "R10_title_from_im": t_from_i[2][1],
           "R1_im_from_title": i_from_t[0][1], "R5_im_from_title": i_from_t[1][1], "R10_im_from_title": i_from_t[2][1]}
    if getattr(dataset, "synthetic", False):
        # beside the reference's six keys (evaluation/eval.py:131-138): the numbers are on synthetic stand-in data
        out["synthetic"] = True
        out["n_pairs"] = len(dataset)
    if rank == 0:
        with open(save_path, "w") as f:
            json.dump(out, f)
    return out, res_vis, res_text


def cli(argv=None):
    ap = argparse.ArgumentParser(description="VTC eval (MI355X)")
    ap.add_argument("-c", "--config", default="configs/pretrained_clip.jsonc", type=str)
    ap.add_argument("-r", "--resume", default=None, type=str)
    ap.add_argument("-d", "--device", default="0", type=str)
    ap.add_argument("--num_irrelevant_comments", default=0, type=int)
    ap.add_argument("--bs", "--batch_size", dest="bs", type=int, default=None)
    ap.add_argument("--bv", "--branch_to_adapt_val", dest="bv", type=str, default=None)
    ap.add_argument("--nc", "--num_comms", dest="nc", type=int, default=None)
    ap.add_argument("--am", "--comment_fusion", dest="am", type=str, default=None)
    ap.add_argument("--ac", "--add_comments", dest="ac", type=str, default=None)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--n_pairs", type=int, default=None, help="synthetic datasets only: number of pairs")
    ap.add_argument("--dtype", type=str, default=None, choices=["bf16", "f16", "f32"],
                    help="operand arithmetic of the towers (default: VTC_COMPUTE_DTYPE, else bf16; the reference is fp32)")
    args = ap.parse_args(argv)
    mods = {"batch_size": args.bs, "arch;args;branch_to_adapt_val": args.bv, "dataset;args;num_comms": args.nc,
            "arch;args;comment_fusion": args.am, "dataset;args;add_comments": args.ac, "dataset;args;n_pairs": args.n_pairs}
    config = ConfigParser.from_file(args.config, resume=args.resume, modification=mods)
    out, res_vis, res_text = main(config, args, config.resume, device="cuda:" + args.device)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps(out))
    import torch.distributed as tdist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
    return out, res_vis, res_text


if __name__ == "__main__":
    cli(sys.argv[1:])
