"""cenet_amd.ops.norm — LayerNorm over the last dimension and train-mode BatchNorm (+ fused activation).
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


# =====================================================================================================
# LayerNorm over the last dim (pvtv2.py:117,124,166,69,221-245)
# =====================================================================================================
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, up_scale=None):
        ctx.up_scale = up_scale
        x = _c(x)
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        y = torch.empty_like(x)
        mean, rstd = _empty((rows,), x), _empty((rows,), x)
        kern.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, Cn, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.refs = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        gp, bp = ctx.refs
        g = _c(g)
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        dx = torch.empty_like(x)
        dg, db = grad_buf(gp), grad_buf(bp)
        if dg is None:  # frozen affine: accumulate into scratch
            dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
        _ln_bwd(g, x, gamma, mean, rstd, dx, dg, db, rows, Cn, up_scale=ctx.up_scale)
        return dx, None, None, None, None


class LinearLNFn(Function):
    """LayerNorm(x W^T + b) for a LONG reduction under few output rows — the spatial-reduction conv (as a Linear layer over patch
    rows, conv2d_tok) and its norm, pvtv2.py:93-95,99-100 — bf16 mode: split-K GEMM into a zero-at-rest fp32 accumulator, then ONE
    kernel (cenet_layernorm_fwd_acc_bf16) adds the bias, rounds, normalises and clears the accumulator, where LinearFn +
    LayerNormFn launch cast_clear_bias and layernorm_fwd.  Backward = LayerNormFn's followed by LinearFn's."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, eps):
        x = _c(x)
        K, N = x.shape[-1], W.shape[0]
        R = x.numel() // K
        shape = x.shape[:-1] + (N,)
        acc = _ZeroWs.take(shape, x)
        kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(kern.wq(W, x), 1, K, kfast=1), acc, R, N, K, scr=N, scc=1,
                  splits=kern.pick_splits(R, N, 1, K // 32), atomic=True)
        xpre, y = _act(shape, x), _act(shape, x)
        mean, rstd = _empty((R,), x), _empty((R,), x)
        kern.layernorm_fwd_acc(acc, b, xpre, gamma, beta, y, mean, rstd, R, N, eps)
        _ZeroWs.release(acc)
        ctx.save_for_backward(x, W, xpre, gamma, mean, rstd)
        ctx.refs = (W, b, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, xpre, gamma, mean, rstd = ctx.saved_tensors
        Wp, bp, gp, btp = ctx.refs
        g = _c(g)
        K, N = x.shape[-1], W.shape[0]
        R = x.numel() // K
        d = torch.empty_like(xpre)
        dg, dbt = grad_buf(gp), grad_buf(btp)
        if dg is None:
            dg, dbt = _zeros((N,), xpre), _zeros((N,), xpre)
        _ln_bwd(g, xpre, gamma, mean, rstd, d, dg, dbt, R, N)
        dW, db = grad_buf(Wp), grad_buf(bp)
        if dW is not None and _wgrad_deferrable(N, K, d, x, K=R):
            _wgrad_defer(d, 0, N, 0, x, 0, K, 0, dW, 0, db, N, K, R, 1, 0)
        elif dW is not None or db is not None:
            with _wgrad_side(d, x):
                if dW is not None:
                    kern.gemm(kern.mat_plain(d, 1, N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R, scr=K, scc=1,
                              splits=kern.pick_splits(N, K, 1, (R + 31) // 32), atomic=True, asum=db)
                else:
                    kern.col_sum(d, db, R, N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            kern.gemm(kern.mat_plain(d, N, 1, kfast=1), kern.mat_plain(kern.wq(Wp, x), K, 1, kfast=0), dx, R, K, N, scr=K, scc=1)
        return dx, None, None, None, None, None


def linear_ln_supported(x, W, b) -> bool:
    K, N = x.shape[-1], W.shape[0]
    R = x.numel() // K
    return bool(_bf(x) and K >= 1024 and N % 4 == 0 and N <= 512 and (b is None or b.data_ptr() % 16 == 0)
                and kern.pick_splits(R, N, 1, K // 32) > 1)


def sr_conv_ln(x, H, Wd, W, b, stride, gamma, beta, eps):
    """LayerNorm(Conv2d(k = stride, no padding)(tokens as a map)) -> tokens (pvtv2.py:93-95,99-100)"""
    x = _c(x)
    B, N, C = x.shape
    k = W.shape[2]
    if (k == stride and W.shape[3] == k and k in (2, 4, 8) and H % k == 0 and Wd % k == 0 and N == H * Wd and W.is_contiguous()):
        xp = PatchTokFn.apply(x, H, Wd, k)
        if linear_ln_supported(xp, W, b):
            return LinearLNFn.apply(xp, W, b, gamma, beta, eps)
    return layernorm(conv2d_tok(x, H, Wd, W, b, stride=stride, pad=0, out_layout="tok"), gamma, beta, eps)


def layernorm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps, getattr(x, "_cenet_bscale", None))


class LayerNormResFn(Function):
    """(LN(x), x): the second output is x itself, routed through this node so that the gradient of the residual connection
    x + f(LN(x)) arrives HERE together with the LayerNorm's own — one kernel writes their sum (pvtv2.py:141-142) instead of
    LN-backward followed by autograd's aten::add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, up_scale=None):
        ctx.up_scale = up_scale
        x = _c(x)
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        y = torch.empty_like(x)
        mean, rstd = _empty((rows,), x), _empty((rows,), x)
        kern.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, Cn, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.refs = (gamma, beta)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_res):
        x, gamma, mean, rstd = ctx.saved_tensors
        gp, bp = ctx.refs
        Cn = x.shape[-1]
        rows = x.numel() // Cn
        if g is None:  # only the residual path carried a gradient
            return g_res, None, None, None, None
        g = _c(g)
        dx = torch.empty_like(x)
        dg, db = grad_buf(gp), grad_buf(bp)
        if dg is None:
            dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
        _ln_bwd(g, x, gamma, mean, rstd, dx, dg, db, rows, Cn, dx_add=_c(g_res), up_scale=ctx.up_scale)
        return dx, None, None, None, None


def layernorm_res(x, gamma, beta, eps):
    """returns (LN(x), x_residual): use x_residual (not x) for the skip connection around the normalised branch"""
    return LayerNormResFn.apply(x, gamma, beta, eps, getattr(x, "_cenet_bscale", None))


# =====================================================================================================
# BatchNorm (+ fused activation) on NCHW or [B,C]  (cfam.py:22-32; blocks.py; nlb.py:81; unet.py:175-197)
# =====================================================================================================
class BatchNormFn(Function):
    """tap: also returns x itself (x_tap).  The residual connection around the normalised branch (cfam.py:365-374: x + ls *
    branch(BN(x))) reads the TAP, so its gradient arrives here and the kernel that writes dx adds it (as LayerNormResFn does for
    the encoder blocks) instead of an aten::add launched by autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, rmean, rvar, nbt, training, eps, act, slope, momentum, tap=False):
        x = _c(x)
        B, Cn = x.shape[:2]
        HW = x.numel() // (B * Cn)
        y = torch.empty_like(x)
        if training:
            mean, var = _empty((Cn,), x), _empty((Cn,), x)
            ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
            kern.bn_train_fwd(x, Cn * HW, y, Cn * HW, ws, mean, var, rmean, rvar, momentum, nbt, eps, weight, bias, act, slope,
                              B, Cn, HW)
        else:
            mean, var = rmean, rvar
            kern.bn_apply(x, Cn * HW, y, Cn * HW, mean, var, eps, weight, bias, act, slope, B, Cn, HW)
        ctx.save_for_backward(x, weight, bias, mean, var)
        ctx.refs = (weight, bias)
        ctx.cfg = (training, eps, act, slope)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, weight, bias, mean, var = ctx.saved_tensors
        wp, bp = ctx.refs
        training, eps, act, slope = ctx.cfg
        if g is None:  # only the tap carried a gradient
            return (g_tap,) + (None,) * 11
        if not training:
            raise RuntimeError("cenet_amd BatchNorm backward is implemented for training mode only")
        g = _c(g)
        if g_tap is not None:
            g_tap = _c(g_tap) if g_tap.dtype == g.dtype else _c(g_tap.to(g.dtype))
        B, Cn = x.shape[:2]
        HW = x.numel() // (B * Cn)
        dx = torch.empty_like(x)
        ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
        dg, db = grad_buf(wp), grad_buf(bp)
        if dg is None:
            dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
        kern.bn_bwd(g, Cn * HW, x, Cn * HW, dx, Cn * HW, mean, var, eps, weight, bias, act, slope, B, Cn, HW, ws, dg, db,
                    dx_add=g_tap)
        return (dx,) + (None,) * 11


def batchnorm(x, weight, bias, rmean, rvar, nbt, training, eps=1e-5, act="none", slope=0.0, momentum=0.1, tap=False):
    """tap=True returns (y, x_tap): hand x_tap (not x) to the residual connection around the normalised branch"""
    return BatchNormFn.apply(x, weight, bias, rmean, rvar, nbt, training, eps, act, slope, momentum, tap)


__all__ = [n for n in dir() if not n.startswith("__")]
