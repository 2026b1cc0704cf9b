"""Checkpoints.

The reference writes `torch.save(net.state_dict(), 'best.pth')` (src/main_acdc.py:272-289) and reads it back with
`net.load_state_dict(torch.load(path, weights_only=True))` (main_acdc.py:152-160); `cenet_amd.networks.CENet` keeps the
reference's 801 state-dict keys, so those files load here unchanged and files written here load in the reference
(`save_weights` / `load_weights`).  The reference keeps no optimizer or scheduler state, so it cannot resume a run;
`save_training_state` / `load_training_state` add that: weights + the SGD momentum arena + step counters + the poly-LR
position, in one file whose `"model"` entry is again a plain reference-compatible state dict.
"""
from __future__ import annotations

from typing import Optional

import torch


def save_weights(net: torch.nn.Module, path: str) -> None:
    """main_acdc.py:278,287: a bare state dict (CPU tensors, so the file opens on any machine)."""
    torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, path)


def load_weights(net: torch.nn.Module, path: str, strict: bool = True):
    """main_acdc.py:157,160.  Accepts a bare state dict or a training-state file written by `save_training_state`."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    return net.load_state_dict(sd, strict=strict)


def save_training_state(path: str, net: torch.nn.Module, optimizer=None, scheduler=None, extra: Optional[dict] = None) -> None:
    state = {"model": {k: v.detach().cpu() for k, v in net.state_dict().items()}}
    if optimizer is not None:
        osd = optimizer.state_dict()
        state["optimizer"] = {"momentum": {n: v.detach().cpu() for n, v in osd["momentum"].items()}, "steps": int(osd["steps"]),
                              "lr": float(osd["lr"])}
    if scheduler is not None:
        state["scheduler"] = {"last_epoch": int(scheduler.last_epoch), "base_lr": float(scheduler.base_lr)}
    if extra:
        state["extra"] = dict(extra)
    torch.save(state, path)


def load_training_state(path: str, net: torch.nn.Module, optimizer=None, scheduler=None) -> dict:
    """Restores weights (in place: the optimizer's parameter arena keeps pointing at them), momentum, counters, LR."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(state["model"], strict=True)
    if optimizer is not None and "optimizer" in state:
        o = state["optimizer"]
        optimizer.load_state_dict(o)
    if scheduler is not None and "scheduler" in state:
        scheduler.last_epoch = state["scheduler"]["last_epoch"]
        scheduler.base_lr = state["scheduler"]["base_lr"]
    return state.get("extra", {})
