"""The adapter-only training entry point (train.py -> vtc_amd/host/train.py): a short run on the synthetic dataset with
a small architecture lowers the loss, writes a checkpoint that the eval entry point loads strictly, and the trained
adapter changes the text embeddings (the towers stay frozen)."""
from dataclasses import asdict

import pytest
import torch

from oracle import arch as A

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_adapter_training_entry_point(tmp_path):
    from vtc_amd.host import train as T
    from vtc_amd.host.clip_arch import ClipConfig
    from vtc_amd.host.parse_config import ConfigParser, read_jsonc
    cfg = read_jsonc("configs/pretrained_clip_comments_attn_frozen.jsonc")
    cfg["arch"]["args"].update(model_type=ClipConfig(**asdict(A.TINY)), n_heads=2)
    cfg["dataset"]["args"].update(n_pairs=96, resolution=A.TINY.image_resolution, context=A.TINY.context_length)
    cfg["batch_size"] = 32
    cfg["trainer"]["epochs"] = 6
    cfg["optimizer"]["args"]["lr"] = 3e-3

    class Args:
        epochs, save = None, str(tmp_path / "adapter.pth")

    config = ConfigParser(cfg)
    model, log, save = T.main(config, Args())
    losses = [e["loss"] for e in log]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0] - 0.05, losses                      # 32-way InfoNCE starts near ln 32 = 3.47
    ck = torch.load(save, map_location="cpu")
    fresh = config.init_obj("arch", __import__("vtc_amd.host.model", fromlist=["x"]))
    fresh.load_state_dict(ck["state_dict"])                           # strict (evaluation/eval.py:90-91)
    before = {k: v.clone() for k, v in fresh.state_dict().items()}
    # towers untouched, adapter moved
    torch.manual_seed(1023)
    ref = config.init_obj("arch", __import__("vtc_amd.host.model", fromlist=["x"]))
    moved = [k for k in before if not torch.equal(before[k], ref.state_dict()[k])]
    assert moved and all(k.startswith("final_transformer.") or k == "mask_embedding" for k in moved), moved[:5]
    # ---- the checkpoint is the reference's (trainer/base_trainer.py:116-145): what its own consumers do with it ----
    assert set(ck) == {"arch", "epoch", "state_dict", "optimizer", "lr_scheduler", "monitor_best", "config"}
    assert ck["arch"] == "PretrainedCLIP_finaltf" and ck["epoch"] == 6
    # evaluation/retrieval_evaluation.py:69 (load_model)
    assert isinstance(ck["config"]["arch"]["args"].get("init_from_avg", False), bool)
    assert ck["config"]["optimizer"]["type"] == "Adam" and ck["config"]["optimizer"]["args"]["lr"] == 3e-3
    assert ck["config"]["lr_scheduler"]["type"] == "StepLR"
    assert ck["monitor_best"] == float("-inf")                       # base_trainer.py:37, monitor "max ..."
    # base_trainer.py:147-215 (_resume_checkpoint): a torch Adam over the reference's parameter groups takes the state
    groups = T.reference_param_groups(fresh, config)
    names = {id(p): n for n, p in fresh.named_parameters()}
    flat = [names[id(p)] for g in groups for p in g["params"]]
    assert flat and all(n.startswith(("final_transformer.", "final_linear.")) or n == "mask_embedding" for n in flat)   # towers frozen
    opt = torch.optim.Adam(groups, **config["optimizer"]["args"])
    opt.load_state_dict(ck["optimizer"])
    n_steps = 6 * (96 // 32)
    stepped = [p for g in opt.param_groups for p in g["params"] if p in opt.state]
    assert stepped and all(float(opt.state[p]["step"]) == n_steps for p in stepped)
    assert all(set(opt.state[p]) >= {"exp_avg", "exp_avg_sq", "max_exp_avg_sq"} for p in stepped)     # amsgrad: true
    assert any(float(opt.state[p]["exp_avg_sq"].abs().sum()) > 0 for p in stepped)
    sched = torch.optim.lr_scheduler.StepLR(opt, **config["lr_scheduler"]["args"])
    sched.load_state_dict(ck["lr_scheduler"])
    assert sched.last_epoch == 6
