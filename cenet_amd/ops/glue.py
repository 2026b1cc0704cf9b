"""cenet_amd.ops.glue — layout / glue operators: token <-> NCHW, concat / split (+ merged depthwise branches), grouped 1x1, add + activation, SiLU product,
residual mix, layer-scale residual.
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


# =====================================================================================================
# layout / glue
# =====================================================================================================
class TokToNCHWFn(Function):
    """pvtv2.py:320-321: [B,N,C] -> [B,C,H,W] contiguous.
    tap: also returns x itself; x's other consumer (the next stage's patch embedding, pvtv2.py:330) reads the tap, and its gradient
    is added by the transpose that writes dx."""

    @staticmethod
    def forward(ctx, x, H, Wd, tap=False):
        x = _c(x)
        B, N, Cn = x.shape
        y = _act((B, Cn, H, Wd), x)
        kern.transpose(x, N * Cn, y, N * Cn, B, N, Cn)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        if g is None:
            return g_tap, None, None, None
        g = _c(g)
        B, Cn, H, Wd = g.shape
        dx = _act((B, H * Wd, Cn), g)
        if g_tap is not None:
            g_tap = _c(g_tap) if g_tap.dtype == g.dtype else _c(g_tap.to(g.dtype))
        kern.transpose(g, Cn * H * Wd, dx, Cn * H * Wd, B, Cn, H * Wd, add=g_tap)
        return dx, None, None, None


def tok_to_nchw(x, H, Wd, tap=False):
    """tap=True returns (x_nchw, x_tap): hand x_tap (not x) to x's other consumer"""
    return TokToNCHWFn.apply(x, H, Wd, tap)


class Concat2Fn(Function):
    """torch.cat([a, b], dim=1) for NCHW (dseb.py:156, out.py:63).
    tap: also returns a and b themselves (a_tap, b_tap); further consumers of a / b that read the TAPS send their gradients through
    this node, where the split kernel adds them to the slices (dseb.py:156-164 + decoders.py:96: `dec` and `skip` each have a
    second consumer) instead of one aten::add per input."""

    @staticmethod
    def forward(ctx, a, b, tap=False):
        a, b = _c(a), _c(b)
        B, Ca = a.shape[:2]
        Cb = b.shape[1]
        HW = a.numel() // (B * Ca)
        y = _act((B, Ca + Cb) + tuple(a.shape[2:]), a)
        kern.cat_channels([a, b], y, B, HW)
        ctx.dims = (Ca, Cb, HW, tuple(a.shape), tuple(b.shape))
        return (y, a.view_as(a), b.view_as(b)) if tap else y

    @staticmethod
    def backward(ctx, g, ga=None, gb=None):
        Ca, Cb, HW, sa, sb = ctx.dims
        if g is None:  # only the taps carried gradients
            return ga, gb, None
        g = _c(g)
        B = g.shape[0]
        da = _act(sa, g)
        db = _act(sb, g)
        ga = _c(ga) if ga is not None and ga.dtype == g.dtype else (None if ga is None else ga.to(g.dtype))
        gb = _c(gb) if gb is not None and gb.dtype == g.dtype else (None if gb is None else gb.to(g.dtype))
        if ga is None and gb is None:
            kern.cat_channels([da, db], g, B, HW, split=True)
        else:
            kern.split_channels_add([da, db], [ga, gb], g, B, HW)
        return da, db, None


def concat2(a, b, tap=False):
    """tap=True returns (cat, a_tap, b_tap): hand the taps (not a / b) to the other consumers of a and b"""
    return Concat2Fn.apply(a, b, tap)


class SplitChannelsFn(Function):
    """x[:, lo:hi] for consecutive channel groups of an NCHW tensor, as contiguous tensors (cfam.py:230)."""

    @staticmethod
    def forward(ctx, x, *sizes):
        x = _c(x)
        B, Cn = x.shape[:2]
        HW = x.numel() // (B * Cn)
        outs, lo = [], 0
        for c in sizes:
            y = _act((B, c) + tuple(x.shape[2:]), x)
            kern.copy_batched(x, Cn * HW, y, c * HW, B, c * HW, x_off=lo * HW)
            outs.append(y)
            lo += c
        ctx.cfg = (tuple(x.shape), sizes, HW)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        shape, sizes, HW = ctx.cfg
        B, Cn = shape[:2]
        ref = next(g for g in gs if g is not None)
        covered = sum(sizes) == Cn and all(g is not None for g in gs)
        dx = _act(shape, ref) if covered else kern.zero_(_act(shape, ref))
        lo = 0
        for c, g in zip(sizes, gs):
            if g is not None:
                kern.copy_batched(_c(g), c * HW, dx, Cn * HW, B, c * HW, y_off=lo * HW)
            lo += c
        return (dx,) + (None,) * len(sizes)


def split_channels(x, sizes):
    return SplitChannelsFn.apply(x, *sizes)


class SplitDWFn(Function):
    """cfam.py:230-236 without the split copies: the first `len(ws)` channel groups of x go straight through their (dilated,
    bias-free) depthwise 3x3 — each kernel reads its slice of x in place and the data-gradient kernels write their slice of
    ONE dx — and the remaining channels come back as a contiguous copy.  Returns (u_0, ..., u_{n-1}, rest)."""

    @staticmethod
    def forward(ctx, x, sizes, dils, joined, *ws):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        outs, lo = [], 0
        used = sum(sizes)
        joint = _act((B, used, H, Wd), x) if joined else None  # joined: the groups' outputs as ONE [B, sum sizes, H, W] tensor
        plan = []  # (x, x_off, sxb, w, y, y_off, syb, C, dil) per branch
        for c, dil, w in zip(sizes, dils, ws):
            if joined:
                plan.append((x, lo * HW, Cn * HW, w, joint, lo * HW, used * HW, c, dil))
            else:
                u = _act((B, c, H, Wd), x)
                plan.append((x, lo * HW, Cn * HW, w, u, 0, c * HW, c, dil))
                outs.append(u)
            lo += c
        # the branches in ONE launch where the library has that form (bf16 planes that fit its LDS tile), else one by one
        if not (_bf(x) and kern.dw_nchw_multi(plan, B, H, Wd, 0)):
            for (xx, xo, sxb, w, y, yo, syb, c, dil) in plan:
                kern.dw_nchw(xx, sxb, w, None, y, syb, None, 0, B, c, H, Wd, dil, 0, x_off=xo, y_off=yo)
        if joined:
            outs = [joint]
        rest = None
        if lo < Cn:
            rest = _act((B, Cn - lo, H, Wd), x)
            kern.copy_batched(x, Cn * HW, rest, (Cn - lo) * HW, B, (Cn - lo) * HW, x_off=lo * HW)
        ctx.save_for_backward(x, *ws)
        ctx.refs = ws
        ctx.cfg = (tuple(sizes), tuple(dils), lo, bool(joined))
        return tuple(outs) + ((rest,) if rest is not None else ())

    @staticmethod
    def backward(ctx, *gs):
        x = ctx.saved_tensors[0]
        ws = ctx.saved_tensors[1:]
        sizes, dils, used, joined = ctx.cfg
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        n = len(sizes)
        nout = 1 if joined else n
        full = all(g is not None for g in gs)
        dx = torch.empty_like(x) if full else kern.zero_(torch.empty_like(x))
        gj = _c(gs[0]) if joined and gs[0] is not None else None
        lo = 0
        dplan, wplan = [], []
        for j, (c, dil, w, wp) in enumerate(zip(sizes, dils, ws, ctx.refs)):
            g = gj if joined else gs[j]
            if g is not None:
                g = _c(g)
                sgb, g_off = (used * HW, lo * HW) if joined else (c * HW, 0)
                dplan.append((g, g_off, sgb, w, dx, lo * HW, Cn * HW, c, dil))
                dw = grad_buf(wp)
                if dw is not None:
                    wplan.append((x, lo * HW, Cn * HW, g, g_off, sgb, dw, c, dil))
            lo += c
        if dplan and not (_bf(x) and kern.dw_nchw_multi(dplan, B, H, Wd, 1)):
            for (g, go, sgb, w, y, yo, syb, c, dil) in dplan:
                kern.dw_nchw(g, sgb, w, None, y, syb, None, 0, B, c, H, Wd, dil, 1, x_off=go, y_off=yo)
        if wplan:
            with _wgrad_side(x, *[b[3] for b in wplan]):
                if not (_bf(x) and kern.dw_wgrad_nchw_multi(wplan, B, H, Wd)):
                    for (xx, xo, sxb, g, go, sgb, dw, c, dil) in wplan:
                        kern.dw_wgrad_nchw(xx, sxb, g, sgb, dw, None, B, c, H, Wd, dil, x_off=xo, g_off=go)
        if used < Cn and len(gs) > nout and gs[nout] is not None:
            kern.copy_batched(_c(gs[nout]), (Cn - used) * HW, dx, Cn * HW, B, (Cn - used) * HW, y_off=used * HW)
        return (dx, None, None, None) + (None,) * n


def split_dwconv(x, sizes, dils, ws, joined=False):
    """joined=True: (U, rest) with U = the groups' outputs side by side in one [B, sum(sizes), H, W] tensor"""
    return SplitDWFn.apply(x, tuple(sizes), tuple(dils), bool(joined), *ws)


class SplitDWBnFn(Function):
    """cfam.py:230-236 over blocks.py:169-177: SplitDWFn(joined) + the branches' (merged) depthwise BatchNorm + ReLU as ONE launch per
    pass (csrc/chanloc.hip: workgroup = channel over the batch; the pooled slice's copy rides along).  -> (V, rest)"""

    @staticmethod
    def forward(ctx, x, sizes, dils, gamma, beta, rmean, rvar, nbt, eps, momentum, *ws):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        g, nb = sizes[0], len(ws)
        p = Cn - g * nb
        v = _act((B, g * nb, H, Wd), x)
        rest = _act((B, p, H, Wd), x) if p > 0 else None
        mean, var = _empty((g * nb,), x), _empty((g * nb,), x)
        kern.dwbn_fwd(x, ws, dils, g, p, v, rest, gamma, beta, eps, mean, var, rmean, rvar, momentum, nbt, B, H, Wd)
        ctx.save_for_backward(x, gamma, beta, mean, var, *ws)
        ctx.refs = (gamma, beta) + tuple(ws)
        ctx.cfg = (g, p, tuple(dils), eps)
        return (v, rest) if rest is not None else (v,)

    @staticmethod
    def backward(ctx, g_v, g_rest=None):
        x, gamma, beta, mean, var = ctx.saved_tensors[:5]
        ws = ctx.saved_tensors[5:]
        g, p, dils, eps = ctx.cfg
        B, Cn, H, Wd = x.shape
        if g_v is None:
            raise RuntimeError("split_dwconv_bn: the branch outputs carried no gradient")
        g_v = _c(g_v)
        if p > 0:
            g_rest = _c(g_rest) if g_rest is not None else kern.zero_(torch.empty((B, p, H, Wd), device=x.device, dtype=x.dtype))
        dx = torch.empty_like(x)
        dws = [_gb(wp, x) for wp in ctx.refs[2:]]
        kern.dwbn_bwd(g_v, g_rest, x, ws, dils, g, p, gamma, beta, eps, mean, var, dx, dws, _gb(ctx.refs[0], x),
                      _gb(ctx.refs[1], x), B, H, Wd)
        return (dx,) + (None,) * (9 + len(ws))


def split_dwconv_bn_supported(x, sizes, training: bool) -> bool:
    return (bool(training) and x.dim() == 4 and len(set(sizes)) == 1 and 1 <= len(sizes) <= 3
            and x.shape[0] * x.shape[2] * x.shape[3] <= 8192 and kern.chanloc_supported(x.shape[0], x.shape[2] * x.shape[3]))


def split_dwconv_bn(x, sizes, dils, ws, gamma, beta, rmean, rvar, nbt, eps, momentum):
    """sizes: the (equal) branch widths; gamma ... nbt: the branches' depthwise BatchNorms joined (ops.merged_param / merged_buffer;
    nbt holds one counter per branch)"""
    return SplitDWBnFn.apply(x, tuple(sizes), tuple(dils), gamma, beta, rmean, rvar, nbt, eps, momentum, *ws)


class GroupedConv1x1Fn(Function):
    """y[b, j*Co + o] = sum_i W[j, o, i] x[b, j*Ci + i]: G independent bias-free 1x1 convolutions on the channel groups of one
    NCHW tensor in one batched GEMM each way (the pointwise convs of the three dilated SepConvBN branches of cfam.py:208-212,
    whose weights ops.merged_param joins into W [G, Co, Ci])."""

    @staticmethod
    def forward(ctx, x, W):
        x = _c(x)
        G, Co, Ci = W.shape[:3]
        B = x.shape[0]
        HW = x.numel() // (B * G * Ci)
        y = _act((B, G * Co) + tuple(x.shape[2:]), x)
        ctx.small = bool(_bf(x) and Co == Ci and kern.pw_small_supported(Ci))
        if ctx.small:  # a handful of channels per group: thread-per-pixel kernel instead of mostly padded GEMM tiles
            kern.pw_small(x, kern.wq(W, x), y, B, G, Ci, HW)
        else:
            kern.gemm(kern.mat_plain(kern.wq(W, x), Ci, 1, sb2=Co * Ci, kfast=1),
                      kern.mat_plain(x, HW, 1, sb=G * Ci * HW, sb2=Ci * HW),
                      y, Co, HW, Ci, scr=HW, scc=1, scb=G * Co * HW, scb2=Co * HW, nbatch=B * G, nb_inner=G)
        ctx.save_for_backward(x, W)
        ctx.refs = (W,)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        Wp, = ctx.refs
        g = _c(g)
        G, Co, Ci = W.shape[:3]
        B = x.shape[0]
        HW = x.numel() // (B * G * Ci)
        dW = grad_buf(Wp)
        if dW is not None:
            with _wgrad_side(g, x):
                iters = B * ((HW + 31) // 32)
                kern.gemm(kern.mat_plain(g, HW, 1, sb=Co * HW, skb=G * Co * HW, kfast=1),
                          kern.mat_plain(x, 1, HW, sb=Ci * HW, skb=G * Ci * HW, kfast=1), dW, Co, Ci, HW, scr=Ci, scc=1,
                          scb=Co * Ci, nbatch=G, nkb=B, splits=kern.pick_splits(Co, Ci, G, iters), atomic=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.small:
                kern.pw_small(g, kern.wq(Wp, x), dx, B, G, Ci, HW, transpose=True)
            else:
                kern.gemm(kern.mat_plain(kern.wq(Wp, x), 1, Ci, sb2=Co * Ci, kfast=0),
                          kern.mat_plain(g, HW, 1, sb=G * Co * HW, sb2=Co * HW),
                          dx, Ci, HW, Co, scr=HW, scc=1, scb=G * Ci * HW, scb2=Ci * HW, nbatch=B * G, nb_inner=G)
        return dx, None


def grouped_conv1x1(x, W):
    return GroupedConv1x1Fn.apply(x, W)


class ConcatFn(Function):
    """torch.cat(xs, dim=1) for NCHW in one pass per input (the nested two-way concats of cfam.py:238 copied twice)."""

    @staticmethod
    def forward(ctx, *xs):
        xs = [_c(t) for t in xs]
        B = xs[0].shape[0]
        cs = [t.shape[1] for t in xs]
        HW = xs[0].numel() // (B * cs[0])
        Ct = sum(cs)
        y = _act((B, Ct) + tuple(xs[0].shape[2:]), xs[0])
        if len(xs) <= 4:
            kern.cat_channels(xs, y, B, HW)  # one launch
        else:
            lo = 0
            for t, c in zip(xs, cs):
                kern.copy_batched(t, c * HW, y, Ct * HW, B, c * HW, y_off=lo * HW)
                lo += c
        ctx.dims = (cs, HW)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        cs, HW = ctx.dims
        B, Ct = g.shape[0], sum(cs)
        outs = [_act((B, c) + tuple(g.shape[2:]), g) for c in cs]
        if len(cs) <= 4:
            kern.cat_channels(outs, g, B, HW, split=True)
        else:
            lo = 0
            for d, c in zip(outs, cs):
                kern.copy_batched(g, Ct * HW, d, c * HW, B, c * HW, x_off=lo * HW)
                lo += c
        return tuple(outs)


def concat(xs):
    return ConcatFn.apply(*xs)


class AddActFn(Function):
    """out = act(a + b) for act in {none, lrelu, relu} (unet.py:212-213; decoders.py:96)."""

    @staticmethod
    def forward(ctx, a, b, act, slope):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        kern.add_act_fwd(a, b, out, a.numel(), act, slope)
        ctx.cfg = (act, slope)
        if act != "none":
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        act, slope = ctx.cfg
        g = _c(g)
        if act == "none":
            return g, g, None, None
        (out,) = ctx.saved_tensors
        d = torch.empty_like(g)
        kern.lrelu_bwd_from_out(out, g, d, g.numel(), slope if act == "lrelu" else 0.0)
        return d, d, None, None


def add_act(a, b, act="none", slope=0.0):
    return AddActFn.apply(a, b, act, slope)


class SiluMulFn(Function):
    """cfam.py:302: SiLU(g) * SiLU(v)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        kern.silu_mul_fwd(a, b, y, a.numel())
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g)
        da, db = torch.empty_like(a), torch.empty_like(b)
        kern.silu_mul_bwd(a, b, g, da, db, a.numel())
        return da, db


def silu_mul(a, b):
    return SiluMulFn.apply(a, b)


class MixFn(Function):
    """nlb.py:147: (1-w) x + w p with a learnable scalar w."""

    @staticmethod
    def forward(ctx, x, p, w):
        x, p = _c(x), _c(p)
        z = torch.empty_like(x)
        kern.mix_fwd(x, p, w, z, x.numel())
        ctx.save_for_backward(x, p, w)
        ctx.refs = (w,)
        return z

    @staticmethod
    def backward(ctx, g):
        x, p, w = ctx.saved_tensors
        g = _c(g)
        dx, dp = torch.empty_like(x), torch.empty_like(p)
        dw = grad_buf(ctx.refs[0])
        if dw is None:
            dw = _zeros((1,), x)
        kern.mix_bwd(x, p, w, g, dx, dp, dw, x.numel())
        return dx, dp, None


def mix(x, p, w):
    return MixFn.apply(x, p, w)


class ScaleResidualFn(Function):
    """cfam.py:368,372: x + layer_scale[c] * y."""

    @staticmethod
    def forward(ctx, x, y, ls):
        x, y = _c(x), _c(y)
        B, Cn = x.shape[:2]
        HW = x.numel() // (B * Cn)
        out = torch.empty_like(x)
        kern.scale_residual_fwd(x, y, ls, out, B, Cn, HW)
        ctx.save_for_backward(y, ls)
        ctx.refs = (ls,)
        return out

    @staticmethod
    def backward(ctx, g):
        y, ls = ctx.saved_tensors
        g = _c(g)
        B, Cn = y.shape[:2]
        HW = y.numel() // (B * Cn)
        dy = torch.empty_like(y)
        kern.scale_chan(g, ls, dy, B, Cn, HW)
        dls = grad_buf(ctx.refs[0])
        if dls is not None:
            kern.chan_dot(g, Cn * HW, y, Cn * HW, dls, B, Cn, HW)
        return g, dy, None


def scale_residual(x, y, ls):
    return ScaleResidualFn.apply(x, y, ls)


__all__ = [n for n in dir() if not n.startswith("__")]
