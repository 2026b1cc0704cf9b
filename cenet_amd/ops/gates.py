"""cenet_amd.ops.gates — CCU and SRM gates (cfam.py:251-264, 93-101).
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


# =====================================================================================================
# CCU and SRM gates (cfam.py:251-264, 93-101)
# =====================================================================================================
class CCUFn(Function):
    """tap: also returns x itself; x's other consumer (the MCA shortcut, cfam.py:298-303) reads the tap and its gradient is added by
    this node's data-gradient kernel"""

    @staticmethod
    def forward(ctx, x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap=False):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        u = _empty((B, Cn, 3), x)
        amax = _empty((B, Cn), x, torch.int32)
        z = _empty((B, Cn), x)
        kern.ccu_stats_fwd(x, fc1, fc2, u, amax, z, B, Cn, HW)
        use_bn = B > 1 and not _Batch1.on
        mean = var = None
        if use_bn:
            zn = torch.empty_like(z)
            if training and kern.bn1d_supported(B):
                # [B, C] fp32 with one value per image and channel: statistics + running update + normalisation in ONE launch
                mean, var = _empty((Cn,), x), _empty((Cn,), x)
                kern.bn1d_train_fwd(z, zn, mean, var, bn_rm, bn_rv, 0.1, bn_nbt, 1e-5, bn_w, bn_b, B, Cn)
            else:
                if training:
                    mean, var = _empty((Cn,), x), _empty((Cn,), x)
                    ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
                    kern.bn_stats(z, Cn, B, Cn, 1, ws, mean, var, bn_rm, bn_rv, 0.1, bn_nbt)
                else:
                    mean, var = bn_rm, bn_rv
                kern.bn_apply(z, Cn, zn, Cn, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, Cn, 1)
        else:
            zn = z
        y = torch.empty_like(x)
        kern.gate_chan_fwd(x, zn, y, B * Cn, HW)
        ctx.save_for_backward(x, fc1, fc2, u, amax, z, zn, mean, var, bn_w, bn_b)
        ctx.refs = (fc1, fc2, bn_w, bn_b)
        ctx.cfg = (use_bn, training)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, fc1, fc2, u, amax, z, zn, mean, var, bn_w, bn_b = ctx.saved_tensors
        use_bn, training = ctx.cfg
        if g is None:
            return (g_tap,) + (None,) * 9
        g = _c(g)
        if g_tap is not None:
            g_tap = _c(g_tap) if g_tap.dtype == g.dtype else _c(g_tap.to(g.dtype))
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        dzn = _empty((B, Cn), x)
        kern.gate_chan_bwd_reduce(x, g, zn, dzn, B * Cn, HW)
        if use_bn:
            if not training:
                raise RuntimeError("CCU backward needs training-mode BatchNorm")
            dz = torch.empty_like(dzn)
            dg, db = grad_buf(ctx.refs[2]), grad_buf(ctx.refs[3])
            if kern.bn1d_supported(B):
                kern.bn1d_bwd(dzn, z, dz, mean, var, 1e-5, bn_w, dg, db, B, Cn)  # (one launch; NULL gradients: frozen affine)
            else:
                ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
                if dg is None:
                    dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
                kern.bn_bwd(dzn, Cn, z, Cn, dz, Cn, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, Cn, 1, ws, dg, db)
        else:
            dz = dzn
        d1, d2 = grad_buf(ctx.refs[0]), grad_buf(ctx.refs[1])
        if d1 is None:
            d1, d2 = _zeros(fc1.shape, x), _zeros(fc2.shape, x)
        dx = torch.empty_like(x)
        kern.ccu_bwd_apply(x, g, zn, dz, u, amax, fc1, fc2, d1, d2, dx, B, Cn, HW, dx_add=g_tap)
        return (dx,) + (None,) * 9


def ccu(x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap=False):
    """tap=True returns (y, x_tap): hand x_tap (not x) to x's other consumer"""
    return CCUFn.apply(x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap)


class SRMFn(Function):
    """cfam.py:93-101.  Training mode on planes of <= 4096 pixels (round 5): the conv + GELU kernel also leaves per-workgroup
    (count, mean, M2) triples, the gate kernel folds them and normalises inline, and backwards ONE kernel does BatchNorm backward,
    GELU' and the conv backward: 3 + 4 launches instead of 6 + 6 (csrc/stats.hip)."""

    @staticmethod
    def forward(ctx, x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        u = _empty((B, 3, H, Wd), x)
        amax = _empty((B, HW), x, torch.int32)
        kern.srm_stats_fwd(x, u, amax, B, Cn, HW)
        f = _empty((B, 1, H, Wd), x)
        fa = torch.empty_like(f)
        fb = torch.empty_like(f)
        y = torch.empty_like(x)
        fused = bool(training) and kern.srm_fused_supported(B, H, Wd)
        if fused:
            G = kern.srm_parts(B, H, Wd)
            part = _empty((G, 3), x)
            mean, var = _empty((1,), x), _empty((1,), x)
            kern.srm_conv_gelu_fwd(u, pwc, dwc, f, fa, part, B, H, Wd)
            kern.gate_pix_bn_fwd(x, fa, part, G, fb, y, bn_w, bn_b, 1e-5, mean, var, bn_rm, bn_rv, 0.1, bn_nbt, B, Cn, HW)
        else:
            kern.srm_conv_fwd(u, pwc, dwc, f, B, H, Wd)
            kern.act_fwd(f, fa, f.numel(), "gelu")
            if training:
                mean, var = _empty((1,), x), _empty((1,), x)
                ws = _empty((2 * 256,), x)  # CENET_BN_WS_FLOATS(1)
                kern.bn_stats(fa, HW, B, 1, HW, ws, mean, var, bn_rm, bn_rv, 0.1, bn_nbt)
            else:
                mean, var = bn_rm, bn_rv
            kern.bn_apply(fa, HW, fb, HW, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, 1, HW)
            kern.gate_pix_fwd(x, fb, y, B, Cn, HW)
        ctx.save_for_backward(x, pwc, dwc, u, amax, f, fa, fb, mean, var, bn_w, bn_b)
        ctx.refs = (pwc, dwc, bn_w, bn_b)
        ctx.training = training
        ctx.fused = fused
        return y

    @staticmethod
    def backward(ctx, g):
        x, pwc, dwc, u, amax, f, fa, fb, mean, var, bn_w, bn_b = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("SRM backward needs training-mode BatchNorm")
        g = _c(g)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        dfb = torch.empty_like(f)
        kern.gate_pix_bwd_reduce(x, g, fb, dfb, B, Cn, HW)
        dg, db = grad_buf(ctx.refs[2]), grad_buf(ctx.refs[3])
        if dg is None:
            dg, db = _zeros((1,), x), _zeros((1,), x)
        du = torch.empty_like(u)
        dp, dd = grad_buf(ctx.refs[0]), grad_buf(ctx.refs[1])
        if dp is None:
            dp, dd = _zeros(pwc.shape, x), _zeros(dwc.shape, x)
        if ctx.fused:
            part2 = _empty((kern.srm_parts(B, H, Wd), 2), x)
            kern.srm_conv_bn_bwd(u, dfb, fa, f, mean, var, 1e-5, bn_w, pwc, dwc, part2, du, dp, dd, dg, db, B, H, Wd)
        else:
            dfa = torch.empty_like(f)
            ws = _empty((2 * 256,), x)  # CENET_BN_WS_FLOATS(1)
            kern.bn_bwd(dfb, HW, fa, HW, dfa, HW, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, 1, HW, ws, dg, db)
            df = torch.empty_like(f)
            kern.act_bwd(f, dfa, df, f.numel(), "gelu")
            kern.srm_conv_bwd(u, df, pwc, dwc, du, dp, dd, B, H, Wd)
        dx = torch.empty_like(x)
        kern.srm_bwd_apply(x, g, fb, u, du, amax, dx, B, Cn, HW)
        return (dx,) + (None,) * 8


def srm(x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training):
    return SRMFn.apply(x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training)


__all__ = [n for n in dir() if not n.startswith("__")]
