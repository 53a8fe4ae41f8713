"""Drop-in for the reference's ``train.py`` restricted to the slice the HIP path implements:
adapter-only fine-tuning -- ``arch.args.freeze`` containing "all" and a ``*_finaltf`` wrapper, i.e.
``configs/pretrained_clip_comments_attn_frozen.jsonc`` -- with ``clip_loss`` and ``torch.optim.Adam`` semantics.

    python train.py -c configs/pretrained_clip_comments_attn_frozen.jsonc [--epochs N] [--bs N] [--n_pairs N] [--save ckpt.pth]

Per batch (train.py:94-192 + trainer/trainer.py's loop): the frozen towers run on the forward path (their outputs
are constants of the step), then ``AdapterTrainer.step`` does the CAM forward, ``clip_loss``, the backward pass and
the Adam update on the GPU.  The ``random_skip_adapter`` draw is made on the host generator exactly where the
reference makes it (model/model.py:163, :200).  At the end the adapter parameters are written back into the wrapper
and a checkpoint ``{"state_dict": ...}`` that ``evaluation/eval.py -r`` accepts is saved.  Anything else the
reference's train.py can do (unfrozen towers, wandb, tensorboard, DataParallel) raises NotImplementedError."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
from torch.utils.data import DataLoader

from . import datasets as module_data
from . import model as module_arch
from .adapter_train import AdapterTrainer
from .parse_config import ConfigParser

EOT = 49407


def main(config: ConfigParser, args=None, device="cuda"):
    arch = config["arch"]
    a = dict(arch.get("args", {}))
    if not arch["type"].endswith("_finaltf") or "all" not in str(a.get("freeze", "")):
        raise NotImplementedError("train: only adapter-only fine-tuning is implemented (a *_finaltf wrapper with freeze='all'); "
                                  "tower backward is outside the hot path (DESIGN.md 7)")
    if config["loss"] != "clip_loss":
        raise NotImplementedError(f"train: loss {config['loss']!r} (only clip_loss, model/loss.py:18-22)")
    oc = config["optimizer"]
    if oc["type"] != "Adam" or float(oc["args"].get("weight_decay", 0)) != 0.0:
        raise NotImplementedError("train: optimizer must be Adam with weight_decay 0 (configs/pretrained_clip_comments_attn_frozen.jsonc)")
    torch.manual_seed(1023)                                   # train.py:34
    if torch.device(device).type == "cuda" and torch.device(device).index is not None:
        torch.cuda.set_device(torch.device(device))
    model = config.init_obj("arch", module_arch).eval().to(device)   # towers: forward path; adapter train-mode semantics live in AdapterTrainer
    if getattr(model, "random_comment_masking", False):
        raise NotImplementedError("random_comment_masking=True (model/model.py:236-246) is not part of the slice")
    if getattr(model, "residual_activation", None) is not None:
        raise NotImplementedError(f"train: residual_activation {model.residual_activation!r} has no backward on the HIP path "
                                  "(the adapter-only step covers residual_activation=None, the shipped config)")
    dataset = config.init_obj("dataset", module_data, train=True, test=False)
    loader = DataLoader(dataset, batch_size=config["batch_size"], shuffle=True, drop_last=True, num_workers=0,
                        generator=torch.Generator().manual_seed(1023))
    lr = float(config.config.get("adapter_lr") or oc["args"].get("lr", 1e-3))
    trainer = AdapterTrainer({k: v for k, v in model.state_dict().items()}, n_layers=len(model.final_transformer.resblocks),
                             n_heads=model.final_transformer.heads, branch=model.branch_to_adapt, lr=lr,
                             betas=tuple(oc["args"].get("betas", (0.9, 0.999))), eps=float(oc["args"].get("eps", 1e-8)),
                             amsgrad=bool(oc["args"].get("amsgrad", False)))
    sched = config.config.get("lr_scheduler", {"type": "StepLR", "args": {"step_size": 1 << 30, "gamma": 1.0}})
    if sched["type"] != "StepLR":
        raise NotImplementedError("train: lr_scheduler must be StepLR")
    epochs = int(getattr(args, "epochs", None) or config["trainer"]["epochs"])
    log = []
    enc_vis = model.encode_image if model._video_tower else model._encode_vis
    with torch.no_grad():
        for epoch in range(epochs):
            trainer.lr = lr * float(sched["args"]["gamma"]) ** (epoch // int(sched["args"]["step_size"]))
            tot, nb = 0.0, 0
            for vis, title, comments, _meta in loader:
                vis, title, comments = vis.to(device), title.to(device), comments.to(device)
                b, nc, ntok = comments.shape
                fv, ft = enc_vis(vis), model.encode_text(title)
                fc = model.encode_text(comments.reshape(b * nc, ntok)).reshape(b, nc, -1).permute(1, 0, 2).contiguous()
                empty = comments[..., 1] == EOT               # model/model.py:208
                skip = None
                if model.random_skip_adapter:
                    torch.rand([])                            # :163 consumes one draw before the mask
                    skip = torch.rand(b) > 0.5                # :200
                loss = trainer.step(fv, ft, fc, empty, skip)
                tot += float(loss)
                nb += 1
            log.append({"epoch": epoch + 1, "loss": tot / max(1, nb), "lr": trainer.lr})
            print(json.dumps(log[-1]), flush=True)
    named = dict(model.named_parameters())
    with torch.no_grad():
        for k, v in trainer.params.items():
            named[k].copy_(v.reshape(named[k].shape))         # in-place (bumps the version: the packed weights are rebuilt)
    save = getattr(args, "save", None) or os.path.join(config["trainer"].get("save_dir", "saved/"), config["name"],
                                                       f"checkpoint-epoch{epochs}.pth")
    os.makedirs(os.path.dirname(os.path.abspath(save)), exist_ok=True)
    torch.save(checkpoint_state(model, trainer, config, epochs, lr), save)
    return model, log, save


def reference_param_groups(model, config):
    """The optimizer parameter groups exactly as train.py:94-192 builds them (rest / final adapter @ adapter_lr /
    CLIP final linears @ fc_lr / time layers @ time_lr, each split into decay / no-decay), so that the saved
    optimizer state carries the parameter numbering the reference's ``optimizer.load_state_dict`` expects."""
    cfg = config.config
    fc_lr, time_lr, adapter_lr = cfg.get("fc_lr"), cfg.get("time_lr"), cfg.get("adapter_lr")
    clip_final_linear = ["model.text_projection", "model.visual.proj"]
    time_layers = ["time", "temporal"]
    final_adapter_layers = ["final_transformer.", "final_linear.", "mask_embedding"]
    nodecay = ["bias", ".ln", "embedding", "temporal_embed"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    lin = [(n, p) for n, p in named if n in clip_final_linear]
    tim = [(n, p) for n, p in named if any(t in n for t in time_layers)]
    ada = [(n, p) for n, p in named if any(t in n for t in final_adapter_layers)]
    taken = {id(p) for _, p in lin + tim + ada}
    rest = [(n, p) for n, p in named if id(p) not in taken]

    def dicts(nps, lr_):
        out = []
        dec = [p for n, p in nps if all(t not in n for t in nodecay)]
        nod = [p for n, p in nps if any(t in n for t in nodecay)]
        for ps, extra in ((dec, {}), (nod, {"weight_decay": 0.0})):
            if ps:
                d = {"params": ps, **extra}
                if lr_ is not None:
                    d["lr"] = lr_
                out.append(d)
        return out

    return dicts(rest, None) + dicts(ada, adapter_lr) + dicts(lin, fc_lr) + dicts(tim, time_lr)


def checkpoint_state(model, trainer, config, epoch, base_lr):
    """The reference's checkpoint (trainer/base_trainer.py:116-145): ``{arch, epoch, state_dict, optimizer,
    lr_scheduler, monitor_best, config}``.  ``config`` is the plain dict (indexable like the reference's ConfigParser:
    evaluation/retrieval_evaluation.py:69 reads ``checkpoint["config"]["arch"]["args"]``, base_trainer.py:161,178-189
    read ``["arch"]``, ``["optimizer"]``, ``["lr_scheduler"]``).  ``optimizer`` / ``lr_scheduler`` are the state dicts of
    REAL torch objects built over the reference's parameter groups, with this trainer's Adam moments (exp_avg,
    exp_avg_sq, max_exp_avg_sq, step) put in, so ``optimizer.load_state_dict`` of the reference resumes them."""
    import warnings
    cfg = config.config
    oc = cfg["optimizer"]
    cpu_model = {n: p for n, p in model.named_parameters()}
    groups = reference_param_groups(model, config)
    opt = getattr(torch.optim, oc["type"])(groups, **oc["args"])
    by_param = {id(p): n for n, p in cpu_model.items()}
    for g in opt.param_groups:
        for p in g["params"]:
            n = by_param[id(p)]
            if n in trainer.params:                            # parameters that received gradients (torch keeps no state for the others)
                st = {"step": torch.tensor(float(trainer.t)), "exp_avg": trainer.m[n].reshape(p.shape).clone(),
                      "exp_avg_sq": trainer.v[n].reshape(p.shape).clone()}
                if trainer.amsgrad:
                    st["max_exp_avg_sq"] = trainer.vmax[n].reshape(p.shape).clone()
                opt.state[p] = st
    sc = cfg.get("lr_scheduler")
    sched_state = None
    if sc is not None:
        sched = getattr(torch.optim.lr_scheduler, sc["type"])(opt, **sc["args"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(epoch):
                sched.step()
        sched_state = sched.state_dict()
    mon = cfg.get("trainer", {}).get("monitor", "off")
    mnt_best = 0 if mon == "off" else (float("inf") if mon.split()[0] == "min" else float("-inf"))   # base_trainer.py:27-37
    osd = opt.state_dict()
    for st in osd["state"].values():
        for k, v in st.items():
            if torch.is_tensor(v):
                st[k] = v.cpu()
    return {"arch": type(model).__name__, "epoch": epoch, "state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
            "optimizer": osd, "lr_scheduler": sched_state, "monitor_best": mnt_best, "config": json.loads(json.dumps(cfg, default=str))}


def cli(argv=None):
    ap = argparse.ArgumentParser(description="VTC adapter-only training (MI355X)")
    ap.add_argument("-c", "--config", default="configs/pretrained_clip_comments_attn_frozen.jsonc", type=str)
    ap.add_argument("-d", "--device", default="0", type=str)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--bs", "--batch_size", dest="bs", type=int, default=None)
    ap.add_argument("--lr", "--learning_rate", dest="lr", type=float, default=None)
    ap.add_argument("--n_pairs", type=int, default=None, help="SyntheticPairs only: number of pairs")
    ap.add_argument("--save", type=str, default=None)
    args = ap.parse_args(argv)
    mods = {"batch_size": args.bs, "optimizer;args;lr": args.lr, "dataset;args;n_pairs": args.n_pairs}
    config = ConfigParser.from_file(args.config, modification=mods)
    return main(config, args, device="cuda:" + args.device)


if __name__ == "__main__":
    cli(sys.argv[1:])
