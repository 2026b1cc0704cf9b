"""cenet_amd.ops.depthwise — depthwise 3x3 convolutions (token layout and NCHW) and the fused PVT Mlp half (csrc/pvt_mlp.hip).
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


# =====================================================================================================
# depthwise 3x3 (+bias, +activation)
# =====================================================================================================
class DWConvTokFn(Function):
    """pvtv2.py:42-43,359-370: GELU(DW3x3(x)+b) on [B, H*W, C] tokens."""

    @staticmethod
    def forward(ctx, x, w, b, H, Wd, act):
        x = _c(x)
        B, N, Cn = x.shape
        # bf16 tokens: the pre-activation is not stored; backward recomputes it from x inside the fused
        # activation-gradient + weight-gradient kernel (dwconv.hip, MODE 2)
        recompute = act != "none" and kern.dw_tok_tiled(x)
        u = torch.empty_like(x) if not recompute else None
        a = torch.empty_like(x) if act != "none" else None
        kern.dw_tok(x, w, b, u, a, B, Cn, H, Wd, 0, act)
        ctx.save_for_backward(x, w, u if (act != "none" and not recompute) else None)
        ctx.refs = (w, b)
        ctx.cfg = (H, Wd, act)
        ctx.recompute = recompute
        return a if a is not None else u

    @staticmethod
    def backward(ctx, g):
        x, w, u = ctx.saved_tensors
        wp, bp = ctx.refs
        H, Wd, act = ctx.cfg
        g = _c(g)
        B, N, Cn = x.shape
        dw, db = grad_buf(wp), grad_buf(bp)
        if ctx.recompute and dw is not None and kern.dw_tok_tiled(g):
            gu = torch.empty_like(g)
            kern.dw_tok_bwd_pre(x, g, w, bp, gu, dw, db, B, Cn, H, Wd, act)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                kern.dw_tok(gu, w, None, dx, None, B, Cn, H, Wd, 1)
            return dx, None, None, None, None, None
        gu = g
        if act != "none":
            if u is None:  # (recompute path without a weight gradient to fuse into: rebuild the pre-activation)
                u = torch.empty_like(x)
                kern.dw_tok(x, w, bp, u, None, B, Cn, H, Wd, 0)
            gu = torch.empty_like(g)
            kern.act_bwd(u, g, gu, g.numel(), act)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            kern.dw_tok(gu, w, None, dx, None, B, Cn, H, Wd, 1)
        if dw is not None:
            with _wgrad_side(gu, x):
                kern.dw_wgrad_tok(x, gu, dw, db, B, Cn, H, Wd)
        return dx, None, None, None, None, None


class PvtMlpFn(Function):
    """pvtv2.py:145-149 (second half) with ONE forward kernel on bf16 tokens: x + s_b * Mlp(LayerNorm(x)), Mlp =
    fc2(GELU(DW3x3(fc1(.)))) (pvtv2.py:40-47, 364-370; csrc/pvt_mlp.hip).  The kernel stores what the backward pass reads (LN
    output + statistics, fc1 output, GELU output) as it goes; the backward pass is the chain of LayerNormResFn / LinearFn /
    DWConvTokFn backward launches, on those tensors."""

    @staticmethod
    def forward(ctx, x, H, Wd, ln_g, ln_b, eps, w1, b1, wd, bd, w2, b2, bscale, up_scale=None):
        ctx.up_scale = up_scale
        x = _c(x)
        B, N, Cn = x.shape
        HD = w1.shape[0]
        y = torch.empty_like(x)
        saved = None
        if any(ctx.needs_input_grad):
            saved = (torch.empty_like(x), _empty((B * N,), x), _empty((B * N,), x),
                     torch.empty((B, N, HD), device=x.device, dtype=x.dtype), torch.empty((B, N, HD), device=x.device, dtype=x.dtype))
        kern.pvt_mlp_fwd(x, ln_g, ln_b, eps, kern.wq(w1, x), b1, wd, bd, kern.wq(w2, x), b2, bscale, y, B, H, Wd, Cn, HD, saved)
        if saved is not None:
            ctx.save_for_backward(x, bscale, *saved)
        ctx.refs = (ln_g, ln_b, w1, b1, wd, bd, w2, b2)
        ctx.cfg = (H, Wd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, bscale, xn, mean, rstd, h, a = ctx.saved_tensors
        ln_g, ln_b, w1, b1, wd, bd, w2, b2 = ctx.refs
        H, Wd = ctx.cfg
        g = _c(g)
        B, N, Cn = x.shape
        HD = w1.shape[0]
        R = B * N

        def wgrad(gy, xin, Wp, bp, Nn, K):
            dW, db = grad_buf(Wp), grad_buf(bp)
            if dW is not None and _wgrad_deferrable(Nn, K, gy, xin, K=R):
                _wgrad_defer(gy, 0, Nn, 0, xin, 0, K, 0, dW, 0, db, Nn, K, R, 1, 0)
            elif dW is not None or db is not None:
                with _wgrad_side(gy, xin):
                    if dW is not None:
                        kern.gemm(kern.mat_plain(gy, 1, Nn, kfast=0), kern.mat_plain(xin, K, 1, kfast=0), dW, Nn, K, R, scr=K,
                                  scc=1, splits=kern.pick_splits(Nn, K, 1, (R + 31) // 32), atomic=True, asum=db)
                    else:
                        kern.col_sum(gy, db, R, Nn)

        # two kernels: (scale + fc2 data gradient + depthwise / GELU backward) and (depthwise data gradient + fc1 data gradient +
        # LayerNorm backward with the residual connection), csrc/pvt_mlp.hip
        dwd, dbd = grad_buf(wd), grad_buf(bd)
        if dwd is None:
            dwd, dbd = _zeros(wd.shape, x), _zeros(bd.shape, x)
        dg, db = grad_buf(ln_g), grad_buf(ln_b)
        if dg is None:
            dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
        gu, dh, dx = torch.empty_like(a), torch.empty_like(a), torch.empty_like(x)
        dxs = None
        if ctx.up_scale is not None and _WgradCfg.prescale and ctx.up_scale.numel() == B:
            # x = residual + up_scale_b * proj(...) (the attention half): its backward wants up_scale_b * dx (_prescaled_put)
            dxs = torch.empty_like(x)
            _prescaled_put(dx, ctx.up_scale, dxs)
        kern.pvt_mlp_bwd(g, bscale, kern.wq(w1, x), kern.wq(w2, x), wd, bd, h, x, ln_g, mean, rstd, gu, dh, dx, dwd, dbd, dg, db,
                         grad_buf(b2), B, H, Wd, Cn, HD, up_scale=ctx.up_scale if dxs is not None else None, dxs=dxs)
        # a was saved as s_b * GELU(.): dW2 = (s_b g)^T a = g^T (s_b a), no scaled copy of g; the bias gradient (column sums of
        # s_b g) comes from the second kernel, so the recorded problem carries no bias
        wgrad(g, a, w2, None, Cn, HD)
        wgrad(dh, xn, w1, b1, HD, Cn)
        return (dx,) + (None,) * 13


def pvt_mlp_supported(x, HD, H, Wd) -> bool:
    return x.dim() == 3 and kern.pvt_mlp_supported(x, x.shape[-1], HD, H, Wd) and os.environ.get("CENET_PVT_MLP_FUSED", "1") != "0"


def pvt_mlp(x, H, Wd, ln_g, ln_b, eps, w1, b1, wd, bd, w2, b2, bscale=None):
    return PvtMlpFn.apply(x, H, Wd, ln_g, ln_b, eps, w1, b1, wd, bd, w2, b2, bscale, getattr(x, "_cenet_bscale", None))


class DWConvNCHWFn(Function):
    """cfam.py:150-151 (bias+GELU), blocks.py:173 (dilated, no bias), blocks.py:305."""

    @staticmethod
    def forward(ctx, x, w, b, dil, act):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        u = torch.empty_like(x)
        a = torch.empty_like(x) if act != "none" else None
        kern.dw_nchw(x, Cn * H * Wd, w, b, u, Cn * H * Wd, a, Cn * H * Wd, B, Cn, H, Wd, dil, 0, act)
        ctx.save_for_backward(x, w, u if act != "none" else None)
        ctx.refs = (w, b)
        ctx.cfg = (dil, act)
        return a if a is not None else u

    @staticmethod
    def backward(ctx, g):
        x, w, u = ctx.saved_tensors
        wp, bp = ctx.refs
        dil, act = ctx.cfg
        g = _c(g)
        B, Cn, H, Wd = x.shape
        sb = Cn * H * Wd
        gu = g
        if act != "none":
            gu = torch.empty_like(g)
            kern.act_bwd(u, g, gu, g.numel(), act)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            kern.dw_nchw(gu, sb, w, None, dx, sb, None, 0, B, Cn, H, Wd, dil, 1)
        dw, db = grad_buf(wp), grad_buf(bp)
        if dw is not None:
            with _wgrad_side(gu, x):
                kern.dw_wgrad_nchw(x, sb, gu, sb, dw, db, B, Cn, H, Wd, dil)
        return dx, None, None, None, None


class DWActFn(Function):
    """cfam.py:150-151: act(DW3x3(x) + bias) as ONE launch per pass at the small decoder levels (csrc/chanloc.hip: workgroup =
    channel over the batch); the pre-activation is recomputed in the backward pass instead of stored."""

    @staticmethod
    def forward(ctx, x, w, b, dil, act):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        y = torch.empty_like(x)
        kern.dwact_fwd(x, w, b, y, act, 0.0, dil, B, Cn, H, Wd)
        ctx.save_for_backward(x, w, b)
        ctx.refs = (w, b)
        ctx.cfg = (dil, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, b = ctx.saved_tensors
        wp, bp = ctx.refs
        dil, act = ctx.cfg
        g = _c(g)
        B, Cn, H, Wd = x.shape
        dx = torch.empty_like(x)
        kern.dwact_bwd(g, x, w, b, dx, _gb(wp, x), grad_buf(bp), act, 0.0, dil, B, Cn, H, Wd)
        return dx, None, None, None, None


def dwconv_tok(x, w, b, H, Wd, act="none"):
    return DWConvTokFn.apply(x, w, b, H, Wd, act)


def dwconv_nchw(x, w, b=None, dil=1, act="none"):
    if act != "none" and x.dim() == 4 and torch.is_grad_enabled() and kern.dwact_supported(x):
        return DWActFn.apply(x, w, b, dil, act)
    return DWConvNCHWFn.apply(x, w, b, dil, act)


__all__ = [n for n in dir() if not n.startswith("__")]
