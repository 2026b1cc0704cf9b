"""cenet_amd.ops.linear — GEMM-backed operators: Linear / MultiLinear in token layout, 1x1 convolutions on NCHW, dense k x k convolutions (implicit GEMM, direct
kernels, token patch rows), the spatial-reduction conv + LayerNorm.
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


# =====================================================================================================
# Linear in token layout (pvtv2.py:41,45,90,98,106; multihead_diffattn.py:79-81,126)
# =====================================================================================================
class LinearFn(Function):
    """y = bscale[b] * (x W^T + b) + resid.  x [..., K] contiguous; bscale [B] needs x [B, n, K].
    tap: also return x itself as a second output; a further consumer of x that reads the TAP instead of x sends its gradient
    through this node, where it rides in the data-gradient GEMM's epilogue (R) instead of an aten::add launched by autograd
    (q beside the spatial-reduction / kv branch of pvtv2.py:97-107, the q/k/v projections of multihead_diffattn.py:79-81)."""

    @staticmethod
    def forward(ctx, x, W, b, resid, bscale, split_k=False, tap=False):
        x = _c(x)
        K = x.shape[-1]
        N = W.shape[0]
        R = x.numel() // K
        M = R  # one flat GEMM over all rows; the per-sample scale is looked up by row (bscale_rows)
        bs_rows = 0 if bscale is None else R // x.shape[0]
        resid = _c(resid)
        shape = x.shape[:-1] + (N,)
        splits = 1
        Wq = kern.wq(W, x)  # fp32 weight, or its bf16 shadow for bf16 activations
        if split_k and bscale is None and K >= 1024 and _bf(x):  # parity mode keeps a deterministic forward
            splits = kern.pick_splits(M, N, 1, K // 32)
        if splits > 1 and resid is None and N % 4 == 0 and (b is None or b.data_ptr() % 16 == 0):
            # few output tiles under a long reduction (the spatial-reduction convs as GEMMs, K = C s^2): split K over workgroups; the
            # partial sums are added atomically into a zero-at-rest fp32 accumulator, and one pass adds the bias, rounds to the
            # activation type and leaves the accumulator zero (no bias pre-fill, no zero fill: 3 launches -> 2)
            acc = _ZeroWs.take(shape, x)
            kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(Wq, 1, K, kfast=1), acc, M, N, K, scr=N, scc=1,
                      splits=splits, atomic=True)
            y = _ZeroWs.give_back_as(acc, x, bias=b)
        elif splits > 1:
            # (general form) ... onto an fp32 output pre-filled with bias + residual, which is then rounded to the activation type
            if resid is not None:
                y = _acc32(shape, x, resid)
                if b is not None:
                    y += b
            elif b is not None:
                y = b.expand(shape).contiguous()
            else:
                y = _zeros(shape, x)
            kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(Wq, 1, K, kfast=1), y, M, N, K, scr=N, scc=1,
                      splits=splits, atomic=True)
            y = kern.cast(y, x.dtype)
        else:
            y = _act(shape, x)
            kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(Wq, 1, K, kfast=1), y, M, N, K, scr=N, scc=1,
                      bias=b, bscale=bscale, bscale_rows=bs_rows, R=resid, srr=N, src=1)
        ctx.save_for_backward(x, W, bscale)
        ctx.refs = (W, b)
        ctx.has_resid = resid is not None
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, W, bscale = ctx.saved_tensors
        Wp, bp = ctx.refs
        if g is None:  # only the tap carried a gradient
            return g_tap, None, None, None, None, None, None
        g = _c(g)
        K, N = x.shape[-1], W.shape[0]
        R = x.numel() // K
        gs = g
        if bscale is not None:
            gs = _prescaled_take(g, bscale)  # (the LayerNorm backward that produced g wrote bscale * g beside it)
            if gs is None:
                gs = torch.empty_like(g)
                kern.scale_batch(g, bscale, gs, x.shape[0], g.numel() // x.shape[0])
        dW, db = grad_buf(Wp), grad_buf(bp)
        if dW is not None and _wgrad_deferrable(N, K, gs, x, K=R):
            # recorded, not launched: reduced with the other weight gradients of the segment by one grouped launch
            _wgrad_defer(gs, 0, N, 0, x, 0, K, 0, dW, 0, db, N, K, R, 1, 0)
        elif dW is not None or db is not None:
            with _wgrad_side(gs, x, returned=(g if ctx.has_resid and gs is g else None)):
                if dW is not None:
                    # the bias gradient (column sums of the output gradient) rides in the weight-gradient pass (asum)
                    iters = (R + 31) // 32
                    kern.gemm(kern.mat_plain(gs, 1, N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R, scr=K, scc=1,
                              splits=kern.pick_splits(N, K, 1, iters), atomic=True, asum=db)
                elif db is not None:
                    kern.col_sum(gs, db, R, N)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if g_tap is not None:
                g_tap = _c(g_tap)
            kern.gemm(kern.mat_plain(gs, N, 1, kfast=1), kern.mat_plain(kern.wq(Wp, x), K, 1, kfast=0), dx, R, K, N, scr=K,
                      scc=1, R=g_tap, srr=K, src=1)
        return dx, None, None, (g if ctx.has_resid else None), None, None, None


def linear(x, W, b=None, resid=None, bscale=None, split_k=False, tap=False):
    """tap=True returns (y, x_tap): hand x_tap (not x) to the other consumers of x"""
    out = LinearFn.apply(x, W, b, resid, bscale, split_k, tap)
    if bscale is not None and resid is not None:
        tag_bscale(out[0] if tap else out, bscale)
    return out


# =====================================================================================================
# 1x1 convolution on NCHW (cfam.py:149,158,299,302; nlb.py:106-115,142; blocks.py:178,320; dseb.py:164)
# =====================================================================================================
class MultiLinearFn(Function):
    """(x W_0^T, x W_1^T, ...) for n equally shaped bias-free weights joined by ops.merged_param into W [n, N, K]: the q / k / v
    projections of multihead_diffattn.py:79-81 as ONE launch per pass — forward and weight gradient as a batched GEMM over
    the n weights, the data gradient as one GEMM with n K-batches (when the incoming gradients sit back to back in memory,
    as DiffAttnHeadsFn returns them; otherwise n GEMMs chained through the residual operand)."""

    @staticmethod
    def forward(ctx, x, W, tap=False):
        x = _c(x)
        n, N, K = W.shape[:3]
        R = x.numel() // K
        Y = _act((n,) + tuple(x.shape[:-1]) + (N,), x)
        kern.gemm(kern.mat_plain(x, K, 1, kfast=1), kern.mat_plain(kern.wq(W, x), 1, K, sb=N * K, kfast=1), Y, R, N, K, scr=N, scc=1,
                  scb=R * N, nbatch=n)
        ctx.save_for_backward(x, W)
        ctx.refs = (W,)
        ctx.tap = tap
        # tap: x itself as a last output; x's other consumers read it and their gradient rides in the data-gradient GEMM (R)
        return tuple(Y[j] for j in range(n)) + ((x.view_as(x),) if tap else ())

    @staticmethod
    def backward(ctx, *gs):
        x, W = ctx.saved_tensors
        Wp, = ctx.refs
        n, N, K = W.shape[:3]
        R = x.numel() // K
        g_tap = None
        if ctx.tap:
            gs, g_tap = gs[:-1], gs[-1]
            if all(g is None for g in gs):
                return g_tap, None, None
            if g_tap is not None:
                g_tap = _c(g_tap) if g_tap.dtype == x.dtype else _c(g_tap.to(x.dtype))
        gs = [_c(g) if g is not None else torch.zeros(x.shape[:-1] + (N,), device=x.device, dtype=x.dtype) for g in gs]
        esz = gs[0].element_size()
        joint = all(g.data_ptr() == gs[0].data_ptr() + j * R * N * esz for j, g in enumerate(gs))
        dW = grad_buf(Wp)
        if dW is not None and _wgrad_deferrable(N, K, x, *gs, K=R):
            for j, g in enumerate(gs):
                _wgrad_defer(g, 0, N, 0, x, 0, K, 0, dW, j * N * K, None, N, K, R, 1, 0)
        elif dW is not None:
            with _wgrad_side(x, *gs):
                iters = (R + 31) // 32
                if joint:
                    kern.gemm(kern.mat_plain(gs[0], 1, N, sb=R * N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R,
                              scr=K, scc=1, scb=N * K, nbatch=n, splits=kern.pick_splits(N, K, n, iters), atomic=True)
                else:
                    for j, g in enumerate(gs):
                        kern.gemm(kern.mat_plain(g, 1, N, kfast=0), kern.mat_plain(x, K, 1, kfast=0), dW, N, K, R, scr=K, scc=1,
                                  splits=kern.pick_splits(N, K, 1, iters), atomic=True, c_offset=j * N * K)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            Wq = kern.wq(Wp, x)
            if joint:
                kern.gemm(kern.mat_plain(gs[0], N, 1, skb=R * N, kfast=1), kern.mat_plain(Wq, K, 1, skb=N * K, kfast=0), dx, R, K, N,
                          scr=K, scc=1, nkb=n, R=g_tap, srr=K, src=1)
            else:
                for j, g in enumerate(gs):
                    kern.gemm(kern.mat_plain(g, N, 1, kfast=1), kern.mat_plain(Wq, K, 1, kfast=0, offset=j * N * K), dx, R, K, N,
                              scr=K, scc=1, R=(dx if j else g_tap), srr=K, src=1)
        elif g_tap is not None:
            dx = g_tap
        return dx, None, None


def multi_linear(x, W, tap=False):
    """W [n, N, K] (ops.merged_param of n bias-free Linear weights): returns the n products x W_j^T (+ x_tap with tap=True: hand
    it, not x, to x's other consumers)"""
    return MultiLinearFn.apply(x, W, tap)


class Conv1x1Fn(Function):
    """y[b] = W x[b] + bias + resid ; x [B, Cin, *spatial] contiguous.  tap: as in LinearFn (theta / phi / g of nlb.py:117-119
    and the gate / value / shortcut consumers in cfam.py read one tensor)."""

    @staticmethod
    def forward(ctx, x, W, b, resid, tap=False):
        x = _c(x)
        B, Cin = x.shape[:2]
        HW = x.numel() // (B * Cin)
        Cout = W.shape[0]
        y = _act((B, Cout) + tuple(x.shape[2:]), x)
        resid = _c(resid)
        # the shortcut 1x1 convolution of the one-channel network input (unet.py conv3): a stencil, not a K = 1 GEMM
        ctx.c1 = bool(_bf(x) and Cin == 1 and b is None and resid is None and x.dim() == 4
                      and kern.conv_c1_supported(1, Cout, 1, 1, 0))
        # a few channels in, as many out, no bias (the pooled branch of cfam.py:213-219): thread-per-pixel kernel
        ctx.small = bool(_bf(x) and not ctx.c1 and Cin == Cout and b is None and resid is None and kern.pw_small_supported(Cin))
        # 64 -> a few channels with bias (the head's last layer, unet.py:200-217): thread-per-pixel kernels (conv_c1.hip)
        ctx.fewout = bool(_bf(x) and not ctx.c1 and not ctx.small and resid is None and not tap
                          and kern.pw_fewout_supported(Cin, Cout))
        if ctx.c1:
            kern.conv_c1_fwd(x, W, y, B, Cout, x.shape[2], x.shape[3], 1)
        elif ctx.small:
            kern.pw_small(x, kern.wq(W, x), y, B, 1, Cin, HW)
        elif ctx.fewout:
            kern.pw_fewout_fwd(x, kern.wq(W, x), b, y, B, Cin, Cout, HW)
        else:
            kern.gemm(kern.mat_plain(kern.wq(W, x), Cin, 1, kfast=1), kern.mat_plain(x, HW, 1, sb=Cin * HW), y, Cout, HW, Cin,
                      scr=HW, scc=1, scb=Cout * HW, nbatch=B, bias=b, bias_on_row=True, R=resid, srb=Cout * HW, srr=HW, src=1)
        ctx.save_for_backward(x, W)
        ctx.refs = (W, b)
        ctx.has_resid = resid is not None
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, W = ctx.saved_tensors
        Wp, bp = ctx.refs
        if g is None:
            return g_tap, None, None, None, None
        g = _c(g)
        B, Cin = x.shape[:2]
        HW = x.numel() // (B * Cin)
        Cout = W.shape[0]
        dW, db = grad_buf(Wp), grad_buf(bp)
        if (dW is not None and not ctx.c1 and not (ctx.fewout and kern.pw_fewout_wgrad_supported(Cin, Cout))
                and _wgrad_deferrable(Cout, Cin, g, x, K=HW, nkb=B)):
            _wgrad_defer(g, 0, HW, Cout * HW, x, 0, HW, Cin * HW, dW, 0, db, Cout, Cin, HW, B, 1)
        elif dW is not None or db is not None:
            with _wgrad_side(g, x, returned=(g if ctx.has_resid else None)):
                if dW is not None and ctx.c1:
                    kern.conv_c1_wgrad(x, g, dW, B, Cout, x.shape[2], x.shape[3], 1)
                elif dW is not None and ctx.fewout and kern.pw_fewout_wgrad_supported(Cin, Cout):
                    kern.pw_fewout_wgrad(x, g, dW, db, B, Cin, Cout, HW)
                elif dW is not None:
                    iters = B * ((HW + 31) // 32)
                    kern.gemm(kern.mat_plain(g, HW, 1, skb=Cout * HW, kfast=1), kern.mat_plain(x, 1, HW, skb=Cin * HW, kfast=1),
                              dW, Cout, Cin, HW, scr=Cin, scc=1, nkb=B, splits=kern.pick_splits(Cout, Cin, 1, iters),
                              atomic=True, asum=db)
                elif db is not None:
                    kern.chan_dot(g, Cout * HW, None, 0, db, B, Cout, HW)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if g_tap is not None:
                g_tap = _c(g_tap)
            if ctx.small and g_tap is None:
                kern.pw_small(g, kern.wq(Wp, x), dx, B, 1, Cin, HW, transpose=True)
            elif ctx.fewout and g_tap is None:
                kern.pw_fewout_dgrad(g, kern.wq(Wp, x), dx, B, Cin, Cout, HW)
            else:
                kern.gemm(kern.mat_plain(kern.wq(Wp, x), 1, Cin, kfast=0), kern.mat_plain(g, HW, 1, sb=Cout * HW), dx, Cin, HW,
                          Cout, scr=HW, scc=1, scb=Cin * HW, nbatch=B, R=g_tap, srb=Cin * HW, srr=HW, src=1)
        elif g_tap is not None:
            dx = g_tap
        return dx, None, None, (g if ctx.has_resid else None), None


def conv1x1(x, W, b=None, resid=None, tap=False):
    """tap=True returns (y, x_tap): hand x_tap (not x) to the other consumers of x"""
    return Conv1x1Fn.apply(x, W, b, resid, tap)


# =====================================================================================================
# dense k x k convolution as implicit GEMM (pvtv2.py:164,67; unet.py:156-197; blocks.py:211)
# =====================================================================================================
class Conv2dFn(Function):
    """Input is addressed x[b*sb + c*sc + y*sy + x*sx] (geom), so NCHW maps, token-layout maps and the zero-stride
    channel broadcast of net.py:55 are all read in place.  Output: 'nchw' [B,Cout,Ho,Wo] or 'tok' [B,Ho*Wo,Cout]."""

    @staticmethod
    def forward(ctx, x, W, b, geom):
        B, Cin, H, Wd, sb, sc, sy, sx, stride, pad, out_layout = geom
        Cout, _, k, _ = W.shape
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (Wd + 2 * pad - k) // stride + 1
        Kd = Cin * k * k
        plain_nchw = (sb == Cin * H * Wd and sc == H * Wd and sy == Wd and sx == 1 and out_layout == "nchw" and b is None)
        ctx.direct = bool(plain_nchw and _bf(x) and kern.conv_direct_supported(Cin, Cout, k, stride, pad))
        ctx.c1 = bool(plain_nchw and _bf(x) and not ctx.direct and kern.conv_c1_supported(Cin, Cout, k, stride, pad))
        if ctx.c1:  # bf16, the two convolutions that read the one-channel network input (conv_c1.hip)
            y = _act((B, Cout, Ho, Wo), x)
            kern.conv_c1_fwd(x, W, y, B, Cout, H, Wd, k)
            ctx.save_for_backward(x, W)
            ctx.refs = (W, b)
            ctx.geom = geom
            ctx.out_hw = (Ho, Wo)
            return y
        if ctx.direct:  # bf16 tensors, output-head convs: LDS-halo direct convolution (conv_direct.hip)
            y = _act((B, Cout, Ho, Wo), x)
            kern.conv_direct(x, W, y, B, Cin, Cout, H, Wd, k, 0)
            ctx.save_for_backward(x, W)
            ctx.refs = (W, b)
            ctx.geom = geom
            ctx.out_hw = (Ho, Wo)
            return y
        Bm = kern.mat_im2col(x, sb=sb, skb=0, sci=sc, sy=sy, sx=sx, KH=k, KW=k, Pw=Wo, Hs=H, Ws=Wd, stride=stride,
                             pad=pad, dil=1, patch_is_row=1, transposed=0, kfast=0)
        shape = (B, Cout, Ho, Wo) if out_layout == "nchw" else (B, Ho * Wo, Cout)
        scr, scc = (Ho * Wo, 1) if out_layout == "nchw" else (1, Cout)
        tiles = B * ((Cout + 63) // 64) * ((Ho * Wo + 63) // 64)
        Wq = kern.wq(W, x)
        if tiles <= 256 and Kd >= 1024 and _bf(x):  # parity mode keeps a deterministic forward
            # few output tiles under a long reduction (the 8x8/4x4/2x2 spatial-reduction convs of pvtv2.py:93-95): split K
            # over workgroups; the partial sums are added atomically onto an fp32 output pre-filled with the bias
            if b is None:
                y = _zeros(shape, x)
            else:
                y = (b.view(1, Cout, 1, 1) if out_layout == "nchw" else b.view(1, 1, Cout)).expand(shape).contiguous()
            kern.gemm(kern.mat_plain(Wq, Kd, 1, kfast=1), Bm, y, Cout, Ho * Wo, Kd, scr=scr, scc=scc, scb=Cout * Ho * Wo,
                      nbatch=B, splits=kern.pick_splits(Cout, Ho * Wo, B, Kd // 32), atomic=True)
            y = kern.cast(y, x.dtype)
        else:
            y = _act(shape, x)
            kern.gemm(kern.mat_plain(Wq, Kd, 1, kfast=1), Bm, y, Cout, Ho * Wo, Kd, scr=scr, scc=scc, scb=Cout * Ho * Wo,
                      nbatch=B, bias=b, bias_on_row=True)
        ctx.save_for_backward(x, W)
        ctx.refs = (W, b)
        ctx.geom = geom
        ctx.out_hw = (Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        Wp, bp = ctx.refs
        B, Cin, H, Wd, sb, sc, sy, sx, stride, pad, out_layout = ctx.geom
        Ho, Wo = ctx.out_hw
        Cout, _, k, _ = W.shape
        Kd = Cin * k * k
        g = _c(g)
        if out_layout == "nchw":
            g_sc, g_sp = Ho * Wo, 1  # channel stride, pixel stride of dY
        else:
            g_sc, g_sp = 1, Cout
        dx = None
        if ctx.needs_input_grad[0]:
            # dX[b][ci][q] = sum_{co,ky,kx} W[co,ci,ky,kx] * dY gathered (transposed map); written with x's strides
            assert sy == Wd * sx, "conv2d data-gradient needs a pixel-linear input layout"
            Wq = kern.wq(Wp, x)
            if getattr(ctx, "direct", False) and _bf(g) and kern.conv_direct_supported(Cout, Cin, k, stride, pad):
                dx = torch.empty_like(x)  # same direct kernel, weights read transposed + flipped
                kern.conv_direct(g, W, dx, B, Cout, Cin, H, Wd, k, 1)
            elif stride == 1:
                # gather form: dX = Wt[Cin, Cout*k*k] x transposed-gather(dY); no atomics, no wasted MACs
                dx = torch.empty_like(x)
                Bm = kern.mat_im2col(g, sb=Cout * Ho * Wo, skb=0, sci=g_sc, sy=Wo * g_sp, sx=g_sp, KH=k, KW=k, Pw=Wd,
                                     Hs=Ho, Ws=Wo, stride=stride, pad=pad, dil=1, patch_is_row=1, transposed=1, kfast=0)
                A = kern.mat_plain(Wq, k * k, 1, kfast=1, kinner=k * k, sk_outer=Cin * k * k)
                kern.gemm(A, Bm, dx, Cin, H * Wd, Cout * k * k, scr=sc, scc=sx, scb=sb, nbatch=B)
            else:
                # strided conv: the gather form would multiply stride^2 - 1 zeros per useful MAC (64x for the 8x8/8
                # SR conv). Dense GEMM dXcol[(ci,ky,kx), p] = W^T dY[:, p] with a col2im scatter epilogue instead.
                overlap = k > stride
                exact = (pad == 0 and H % stride == 0 and Wd % stride == 0 and k == stride)
                # overlapping patches are scatter-ADDED (atomics): that needs an fp32 image, rounded to the activation type after
                if overlap:
                    dx = _zeros(x.shape, x)
                else:
                    dx = torch.empty_like(x) if exact else kern.zero_(torch.empty_like(x))
                kern.gemm(kern.mat_plain(Wq, 1, Kd, kfast=0), kern.mat_plain(g, g_sc, g_sp, sb=Cout * Ho * Wo,
                                                                               kfast=int(g_sc == 1)),
                          dx, Kd, Ho * Wo, Cout, scr=0, scc=0, scb=sb, nbatch=B, atomic=overlap,
                          col2im=dict(KH=k, KW=k, Pw=Wo, Hs=H, Ws=Wd, stride=stride, pad=pad, sci=sc, sy=sy, sx=sx))
                dx = kern.cast(dx, x.dtype)
        # a DERIVED weight (not a leaf: the channel-summed stem weight of OverlapPatchEmbed) has no gradient buffer of its own:
        # its gradient is computed into a fresh fp32 tensor on this stream and handed to autograd
        derived = Wp is not None and not Wp.is_leaf
        if derived:
            dW = _zeros(Wp.shape, g) if ctx.needs_input_grad[1] else None
        else:
            dW = grad_buf(Wp)
        db = grad_buf(bp)
        if dW is not None or db is not None:
            with (contextlib.nullcontext() if derived else _wgrad_side(g, x)):
                if dW is not None and getattr(ctx, "c1", False) and _bf(g):
                    kern.conv_c1_wgrad(x, g, dW, B, Cout, H, Wd, k)
                elif (dW is not None and getattr(ctx, "direct", False) and _bf(g)
                        and kern.conv_wgrad_direct_supported(Cin, Cout, k, stride, pad)):
                    kern.conv_wgrad_direct(x, g, dW, B, Cin, Cout, H, Wd, k)  # conv_direct.hip, direct weight gradient
                elif dW is not None:
                    Bm = kern.mat_im2col(x, sb=0, skb=sb, sci=sc, sy=sy, sx=sx, KH=k, KW=k, Pw=Wo, Hs=H, Ws=Wd, stride=stride,
                                         pad=pad, dil=1, patch_is_row=0, transposed=0, kfast=1)
                    iters = B * ((Ho * Wo + 31) // 32)
                    kern.gemm(kern.mat_plain(g, g_sc, g_sp, skb=Cout * Ho * Wo, kfast=int(g_sp == 1)), Bm, dW, Cout, Kd,
                              Ho * Wo, scr=Kd, scc=1, nkb=B, splits=kern.pick_splits(Cout, Kd, 1, iters), atomic=True)
                if db is not None:
                    if out_layout == "nchw":
                        kern.chan_dot(g, Cout * Ho * Wo, None, 0, db, B, Cout, Ho * Wo)
                    else:
                        kern.col_sum(g, db, B * Ho * Wo, Cout)
        return dx, (dW if derived else None), None, None


class ChanSumWeightFn(Function):
    """W [Cout, Cin, k, k] -> sum over Cin [Cout, 1, k, k]: the weight of a conv whose Cin input channels are copies of one channel
    (net.py:55).  Backward ADDS the broadcast gradient into W's gradient buffer, like every weight-gradient kernel of this file (and
    returns nothing to autograd: no AccumulateGrad node, whose stream bookkeeping does not fit a step captured on a side stream)."""

    @staticmethod
    def forward(ctx, W):
        ctx.ref = W
        return W.sum(1, keepdim=True)

    @staticmethod
    def backward(ctx, g):
        dW = grad_buf(ctx.ref)
        if dW is not None:
            dW.add_(g)
        return None


def chan_sum_weight(W):
    return ChanSumWeightFn.apply(W)


def conv2d_nchw(x, W, b=None, stride=1, pad=0, out_layout="nchw", expand_channels: int = 0):
    """x [B,C,H,W] contiguous NCHW.  expand_channels=3 reads a 1-channel input as 3 identical channels (net.py:55)."""
    x = _c(x)
    B, C, H, Wd = x.shape
    if expand_channels and C == 1:
        geom = (B, expand_channels, H, Wd, H * Wd, 0, Wd, 1, stride, pad, out_layout)
    else:
        geom = (B, C, H, Wd, C * H * Wd, H * Wd, Wd, 1, stride, pad, out_layout)
    return Conv2dFn.apply(x, W, b, geom)


class PatchTokFn(Function):
    """tokens [B, H*W, C] -> patch rows [B, (H/s)*(W/s), C*s*s], k = (c, ky, kx) (cenet_patch_tok_f32); the backward is
    the inverse scatter, which writes every input-gradient element exactly once."""

    @staticmethod
    def forward(ctx, x, H, Wd, s):
        x = _c(x)
        B, N, C = x.shape
        Ho, Wo = H // s, Wd // s
        xp = _act((B, Ho * Wo, C * s * s), x)
        kern.patch_tok(x, xp, B, Ho, Wo, C, s)
        ctx.geom = (B, Ho, Wo, C, s, N)
        return xp

    @staticmethod
    def backward(ctx, g):
        B, Ho, Wo, C, s, N = ctx.geom
        g = _c(g)
        dx = _act((B, N, C), g)
        kern.patch_tok(g, dx, B, Ho, Wo, C, s, inverse=True)
        return dx, None, None, None


class Im2colTokFn(Function):
    """tokens [B, H*W, C] -> overlapping K x K patch rows [B, Ho*Wo, C*K*K] (cenet_im2col_tok); backward = the gathering
    transpose, so the convolution's data gradient needs no atomics."""

    @staticmethod
    def forward(ctx, x, H, Wd, K, stride, pad):
        x = _c(x)
        B, N, C = x.shape
        Ho, Wo = (H + 2 * pad - K) // stride + 1, (Wd + 2 * pad - K) // stride + 1
        xp = _act((B, Ho * Wo, C * K * K), x)
        kern.im2col_tok(x, xp, B, H, Wd, C, K, stride, pad)
        ctx.geom = (B, H, Wd, C, K, stride, pad, N)
        return xp

    @staticmethod
    def backward(ctx, g):
        B, H, Wd, C, K, stride, pad, N = ctx.geom
        g = _c(g)
        dx = _act((B, N, C), g)
        kern.im2col_tok(g, dx, B, H, Wd, C, K, stride, pad, inverse=True)
        return dx, None, None, None, None, None


def conv2d_tok(x, H, Wd, W, b=None, stride=1, pad=0, out_layout="tok"):
    """x [B, H*W, C] token layout read as an NCHW map (pvtv2.py:93-94)."""
    x = _c(x)
    B, N, C = x.shape
    k = W.shape[2]
    if (out_layout == "tok" and k == stride and W.shape[3] == k and pad == 0 and k in (2, 4, 8) and H % k == 0
            and Wd % k == 0 and N == H * Wd and W.is_contiguous()):
        # non-overlapping patches (the spatial-reduction conv): gather the patches once, then it is a Linear layer over
        # rows of C*k*k with the weight in its own [Cout, (c, ky, kx)] order -- both GEMM operands k-contiguous
        return linear(PatchTokFn.apply(x, H, Wd, k), W, b, split_k=True)
    if (out_layout == "tok" and _bf(x) and k == 3 and W.shape[3] == 3 and N == H * Wd and W.is_contiguous()):
        # bf16, overlapping 3x3 patches (patch embeddings of stages 2-4): materialise the rows (2.25x the map), then forward,
        # weight gradient and data gradient are plain GEMMs for the LDS-DMA ring kernel
        return linear(Im2colTokFn.apply(x, H, Wd, 3, stride, pad), W, b)
    geom = (B, C, H, Wd, N * C, 1, Wd * C, C, stride, pad, out_layout)
    return Conv2dFn.apply(x, W, b, geom)


__all__ = [n for n in dir() if not n.startswith("__")]
