"""Checkpoint files in the reference's format (model/runner.py:955-1075), so that runs can move between the two
implementations: a ``torch.save``d dict with ``epoch``, ``global_step``, ``stats`` and ``model`` (the LiDAR4D state dict:
same keys and shapes, tests/test_gpu_model.py::test_state_dict_roundtrip), plus -- for "full" checkpoints --
``optimizer`` (torch.optim.Adam layout over ``LiDAR4D.get_params``), ``lr_scheduler`` (LambdaLR), ``scaler`` (GradScaler)
and ``ema`` (torch_ema layout).

Host-side bookkeeping only; nothing here launches a kernel.
"""
import glob
import os

import torch


def scheduler_state(opt):
    """State dict of the reference's ``LambdaLR(optimizer, lambda it: 0.1 ** min(it / iters, 1))`` at FlatAdam's step
    count (main_lidar4d.py:303-305): what torch writes for a lambda scheduler (the lambda itself is not pickled)."""
    n_groups = len(opt.model.get_params(opt.lr0))
    base = [g["lr"] for g in opt.model.get_params(opt.lr0)]
    decay = 0.1 ** min(opt.step_count / opt.iters, 1.0)
    return {"base_lrs": base, "last_epoch": opt.step_count, "verbose": False, "_step_count": opt.step_count + 1,
            "_get_lr_called_within_step": False, "_last_lr": [b * decay for b in base], "lr_lambdas": [None] * n_groups}


def save_checkpoint(path, model, opt=None, ema=None, scaler=None, epoch=0, global_step=0, stats=None, full=True):
    """runner.py:955-977.  ``opt``: trainer.FlatAdam, ``ema``: trainer.FlatEMA, ``scaler``: a torch GradScaler or None
    (the fused path carries its own constant loss scale; an empty dict is stored then, as a disabled GradScaler does)."""
    state = {"epoch": epoch, "global_step": global_step,
             "stats": stats if stats is not None else {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}}
    if full:
        if opt is not None:
            state["optimizer"] = opt.state_dict()
            state["lr_scheduler"] = scheduler_state(opt)
        state["scaler"] = scaler.state_dict() if scaler is not None else {}
        if ema is not None:
            state["ema"] = ema.state_dict()
    state["model"] = model.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(state, path)
    return path


def latest_checkpoint(ckpt_dir, name):
    """runner.py:1016-1019: the newest ``<name>_ep*.pth`` of a workspace, or None."""
    found = sorted(glob.glob(os.path.join(ckpt_dir, f"{name}_ep*.pth")))
    return found[-1] if found else None


def load_checkpoint(path, model, opt=None, ema=None, scaler=None, model_only=False, map_location=None):
    """runner.py:1014-1075.  Accepts a bare state dict (``"model"`` key absent) or a checkpoint dict of either
    implementation; returns the bookkeeping entries {"epoch", "global_step", "stats", "missing_keys", "unexpected_keys"}.
    Loading writes through the flat parameter arena (parameters are views of it), so the fp16 compute copies are
    refreshed on the next forward."""
    ckpt = torch.load(path, map_location=map_location if map_location is not None else next(model.parameters()).device,
                      weights_only=False)
    info = {"epoch": 0, "global_step": 0, "stats": None, "missing_keys": [], "unexpected_keys": []}
    if "model" not in ckpt:
        model.load_state_dict(ckpt)
        return info
    res = model.load_state_dict(ckpt["model"], strict=False)
    info["missing_keys"], info["unexpected_keys"] = list(res.missing_keys), list(res.unexpected_keys)
    if ema is not None and "ema" in ckpt:
        ema.load_state_dict(ckpt["ema"])
    if model_only:
        return info
    for k in ("stats", "epoch", "global_step"):
        if k in ckpt:
            info[k] = ckpt[k]
    if opt is not None and "optimizer" in ckpt:
        opt.load_state_dict(ckpt["optimizer"])
        if "lr_scheduler" in ckpt:  # iterations of the LambdaLR (stepped every iteration, also when the scaler skipped)
            opt.step_count = int(ckpt["lr_scheduler"]["last_epoch"])
            opt.sync_device_schedule()
    if scaler is not None and ckpt.get("scaler"):
        scaler.load_state_dict(ckpt["scaler"])
    return info
