"""Loss interface mirroring reference src/utils/core.py:161-188 (`Criterion(num_classes, args)(outputs, labels)`).

Dice, cross-entropy and BoundaryDoULoss (core.py:44-131) are terms of ONE fused HIP kernel each way
(cenet_amd.ops.dice_ce_loss): the per-class sums they share are accumulated once, the boundary-pixel counts of
BoundaryDoULoss come from the same pass over the labels, and the backward is a single kernel for any weighting.
"""
from __future__ import annotations

import torch.nn as nn

from . import ops


class DiceLoss(nn.Module):
    """core.py:44-80 with softmax=True, weight=None."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        if not softmax or weight is not None:
            raise NotImplementedError("the fused kernel implements DiceLoss(softmax=True, weight=None)")
        return ops.dice_ce_loss(inputs, target, 1.0, 0.0)


class BoundaryDoULoss(nn.Module):
    """core.py:83-131 (no hard-coded .cuda(): the boundary counts are computed on the device that holds the labels)."""

    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target):
        return ops.dice_ce_loss(inputs, target, 0.0, 0.0, 1.0)


class Criterion(nn.Module):
    def __init__(self, num_classes, args):
        super().__init__()
        self.num_classes = num_classes
        names = args.loss_type.split(',')
        weights = [float(w) for w in args.loss_weights.split(',')]
        self.w_dice = self.w_ce = self.w_bd = 0.0
        for n, w in zip(names, weights):
            if n == "dice":
                self.w_dice += w
            elif n == "ce":
                self.w_ce += w
            elif n == "boundary":
                self.w_bd += w
            else:
                raise NotImplementedError(f"Loss {n} not implemented")

    def forward(self, outputs, labels):
        return ops.dice_ce_loss(outputs, labels, self.w_dice, self.w_ce, self.w_bd)
