"""timm.layers stand-in: DropPath, to_2tuple, trunc_normal_, trunc_normal_tf_ (oracle tooling only)."""
import collections.abc
from itertools import repeat

import torch
from torch import nn


class DropPath(nn.Module):
    """Per-sample stochastic depth: x * bernoulli(keep) / keep in training, identity otherwise."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def to_2tuple(v):
    if isinstance(v, collections.abc.Iterable) and not isinstance(v, str):
        return tuple(v)
    return tuple(repeat(v, 2))


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    with torch.no_grad():
        return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def trunc_normal_tf_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    with torch.no_grad():
        nn.init.trunc_normal_(tensor, 0.0, 1.0, a, b)
        tensor.mul_(std).add_(mean)
    return tensor
