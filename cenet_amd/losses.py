"""Loss interface mirroring reference src/utils/core.py:161-188 (`Criterion(num_classes, args)(outputs, labels)`).

The Dice + cross-entropy pair (the hot-path loss of BASELINE.json) runs as one fused HIP kernel each way
(cenet_amd.ops.dice_ce_loss); BoundaryDoULoss is SURVEY.md §8f "next" and raises NotImplementedError.
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


class Criterion(nn.Module):
    def __init__(self, num_classes, args):
        super().__init__()
        self.num_classes = num_classes
        names = args.loss_type.split(',')
        weights = [float(w) for w in args.loss_weights.split(',')]
        self.w_dice = self.w_ce = 0.0
        for n, w in zip(names, weights):
            if n == "dice":
                self.w_dice += w
            elif n == "ce":
                self.w_ce += w
            elif n == "boundary":
                raise NotImplementedError("BoundaryDoULoss is not on the round-1 hot path (SURVEY.md §8f)")
            else:
                raise NotImplementedError(f"Loss {n} not implemented")

    def forward(self, outputs, labels):
        return ops.dice_ce_loss(outputs, labels, self.w_dice, self.w_ce)
