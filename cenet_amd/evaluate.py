"""Evaluation path on the device: batched inference, argmax masks and Dice without leaving HBM.

Mirrors the reference's call sites:
  * `val()` of src/main_acdc.py:218-231 — per batch `dc(argmax(softmax(net(x))), label)` with medpy's binary Dice (non-zero
    = foreground) averaged over the loader  -> `val_dice`, `validate`
  * `calculate_dice_percase` / `test_single_volume` of src/utils/metrics_eval.py:24-34,37-84 — per foreground class over a
    whole volume, with the wrapper rules (pred>0 & gt==0 -> 1, otherwise 0)  -> `volume_class_dice`
The reference predicts slice by slice with batch size 1 and computes the metric on the host; here a volume's slices go
through the network in batches and one kernel produces the masks and the overlap counts (loss_optim.hip).  HD95 / ASD
(medpy surface distances) and the scipy cubic `zoom` of non-224 slices stay host-side and are not part of this module:
slices must already have the network's input size.
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch

from . import kern


def predict_counts(logits: torch.Tensor, labels: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits [B,K,H,W] -> (pred [B,H,W] class ids as float, counts [K+1,3] int32 or None).
    counts[c] = (|pred==c & gt==c|, |pred==c|, |gt==c|); counts[K] = the same for the binary masks pred>0 / gt>0."""
    logits = logits.contiguous()
    B, K = logits.shape[:2]
    HW = logits.numel() // (B * K)
    pred = torch.empty((B,) + tuple(logits.shape[2:]), device=logits.device, dtype=torch.float32)
    counts = None
    if labels is not None:
        labels = labels.reshape(B, -1).contiguous().float()
        counts = torch.empty((K + 1, 3), device=logits.device, dtype=torch.int32)
    kern.argmax_counts(logits, labels, pred, counts, B, K, HW)
    return pred, counts


def binary_dice(counts: torch.Tensor) -> float:
    """medpy.metric.binary.dc on the non-zero masks (main_acdc.py:228): 2|A∩B| / (|A|+|B|), 0.0 when both are empty."""
    inter, npred, ngt = (int(v) for v in counts[-1].tolist())
    return 2.0 * inter / (npred + ngt) if (npred + ngt) else 0.0


def class_dice(counts: torch.Tensor) -> List[float]:
    """calculate_dice_percase (metrics_eval.py:24-34) for classes 1..K-1."""
    out = []
    for inter, npred, ngt in counts[1:-1].tolist():
        if npred > 0 and ngt > 0:
            out.append(2.0 * inter / (npred + ngt))
        elif npred > 0 and ngt == 0:
            out.append(1.0)
        else:
            out.append(0.0)
    return out


@torch.no_grad()
def val_dice(net, images: torch.Tensor, labels: torch.Tensor) -> float:
    """one iteration of val() (main_acdc.py:222-228): images [B,Cin,H,W], labels [B,H,W] (or [B,1,H,W])."""
    _, counts = predict_counts(net(images), labels)
    return binary_dice(counts)


@torch.no_grad()
def validate(net, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]]) -> float:
    """val() of main_acdc.py:218-231: mean over the loader of the per-batch binary Dice; sets eval mode like the reference."""
    net.eval()
    vals = [val_dice(net, x, y) for x, y in batches]
    return sum(vals) / max(len(vals), 1)


@torch.no_grad()
def volume_class_dice(net, volume: torch.Tensor, label: torch.Tensor, classes: int, batch_slices: int = 32) -> List[float]:
    """test_single_volume (metrics_eval.py:37-84) for slices that already have the network input size: volume [D,H,W],
    label [D,H,W]; returns the Dice of classes 1..classes-1 over the whole volume."""
    net.eval()
    total = torch.zeros((classes + 1, 3), device=volume.device, dtype=torch.int64)
    for s in range(0, volume.shape[0], batch_slices):
        x = volume[s:s + batch_slices].unsqueeze(1).float()
        _, counts = predict_counts(net(x), label[s:s + batch_slices])
        total += counts.long()
    return class_dice(total)
