"""cenet_amd.ops.dseb_loss — DSEB combine (dseb.py:40-50,63-76,156-163) and the fused segmentation loss (utils/core.py:44-131,161-188).
Part of the cenet_amd.ops package (split by operator family in round 6; `from cenet_amd import ops` exposes every name as before)."""
from __future__ import annotations

import contextlib
import math
import os
from typing import Optional, Sequence

import torch
from torch.autograd import Function

from .. import kern
from .infra import *  # noqa: F401,F403
from .linear import *  # noqa: F401,F403
from .norm import *  # noqa: F401,F403
from .depthwise import *  # noqa: F401,F403
from .attention import *  # noqa: F401,F403
from .glue import *  # noqa: F401,F403
from .decoder_fused import *  # noqa: F401,F403
from .gates import *  # noqa: F401,F403


# =====================================================================================================
# DSEB combine (dseb.py:40-50,63-76,156-163)
# =====================================================================================================
class DsebCombineFn(Function):
    """z = ycoef*y + w[c]*edge(y, recon_s) + diff*y; `recons[s]` is None for scale 1.0 (e_s == 0); diff may be None."""

    @staticmethod
    def forward(ctx, y, w, diff, ycoef, n, *recons):
        y, diff = _c(y), _c(diff)
        recons = [_c(r) for r in recons]
        B, Cn = y.shape[:2]
        HW = y.numel() // (B * Cn)
        z = torch.empty_like(y)
        kern.dseb_combine_fwd(y, recons, n, w, diff, ycoef, z, B, Cn, HW)
        ctx.save_for_backward(y, w, diff, *[r for r in recons if r is not None])
        ctx.ycoef = ycoef
        ctx.mask = [r is not None for r in recons]
        ctx.refs = (w,)
        ctx.n = n
        return z

    @staticmethod
    def backward(ctx, g):
        y, w, diff = ctx.saved_tensors[:3]
        present = list(ctx.saved_tensors[3:])
        recons, it = [], iter(present)
        for m in ctx.mask:
            recons.append(next(it) if m else None)
        g = _c(g)
        B, Cn = y.shape[:2]
        HW = y.numel() // (B * Cn)
        dy = torch.empty_like(y)
        ddiff = torch.empty_like(y) if diff is not None else None
        drs = [torch.empty_like(y) if r is not None else None for r in recons]
        dw = grad_buf(ctx.refs[0])
        if dw is None:
            dw = _zeros(w.shape, y)
        kern.dseb_combine_bwd(y, recons, ctx.n, w, diff, ctx.ycoef, g, dy, drs, ddiff, dw, B, Cn, HW)
        return (dy, None, ddiff, None, None) + tuple(drs)


def dseb_combine(y, w, diff, recons: Sequence[Optional[Tensor]], ycoef: float = 2.0):
    return DsebCombineFn.apply(y, w, diff, ycoef, len(recons), *recons)


# =====================================================================================================
# loss (utils/core.py:44-80,161-188)
# =====================================================================================================
class DiceCELossFn(Function):
    """w_dice * Dice + w_ce * CE + w_bd * BoundaryDoU in one pass over the logits each way (core.py:44-131,161-188)."""

    @staticmethod
    def forward(ctx, logits, labels, w_dice, w_ce, w_bd=0.0):
        logits, labels = _c(logits), _c(labels)
        B, K = logits.shape[:2]
        H, W = (logits.shape[2], logits.shape[3]) if logits.dim() == 4 else (1, logits.numel() // (B * K))
        acc = _empty((16384,), logits)  # CENET_LOSS_ACC_FLOATS (include/cenet_hip.h): replicated partial sums, one value per line
        loss = _empty((1,), logits)
        kern.seg_loss_fwd(logits, labels, acc, loss, B, K, H, W, w_dice, w_ce, w_bd)
        ctx.save_for_backward(logits, labels, acc)
        ctx.cfg = (w_dice, w_ce, w_bd, H, W)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, labels, acc = ctx.saved_tensors
        w_dice, w_ce, w_bd, H, W = ctx.cfg
        B, K = logits.shape[:2]
        g = _c(g).reshape(1)
        d = torch.empty_like(logits)
        kern.seg_loss_bwd(logits, labels, acc, g, d, B, K, H, W, w_dice, w_ce, w_bd)
        return d, None, None, None, None


def dice_ce_loss(logits, labels, w_dice=0.5, w_ce=0.5, w_boundary=0.0):
    return DiceCELossFn.apply(logits, labels, w_dice, w_ce, w_boundary)


__all__ = [n for n in dir() if not n.startswith("__")]
