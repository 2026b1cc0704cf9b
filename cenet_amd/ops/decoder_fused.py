"""cenet_amd.ops.decoder_fused — resampling operators and the channel-local fused chains of the decoder and the head (csrc/chanloc.hip, csrc/res_tail.hip): EUCB front,
CFAM front / mid, pooled branch, residual-block tails.
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


# =====================================================================================================
# resampling
# =====================================================================================================
class BilinearFn(Function):
    """tap: also returns x itself; x's other consumers read the tap and their gradient is added by the kernel that writes dx"""

    @staticmethod
    def forward(ctx, x, Ho, Wo, sh, sw, align, tap=False):
        x = _c(x)
        B, Cn, Hi, Wi = x.shape
        y = _act((B, Cn, Ho, Wo), x)
        kern.bilinear_fwd(x, Cn * Hi * Wi, y, Cn * Ho * Wo, B, Cn, Hi, Wi, Ho, Wo, sh, sw, align)
        ctx.cfg = (Hi, Wi, Ho, Wo, sh, sw, align)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        Hi, Wi, Ho, Wo, sh, sw, align = ctx.cfg
        if g is None:
            return (g_tap,) + (None,) * 6
        g = _c(g)
        B, Cn = g.shape[:2]
        dx = _act((B, Cn, Hi, Wi), g)
        if g_tap is not None:
            g_tap = _c(g_tap) if g_tap.dtype == g.dtype else _c(g_tap.to(g.dtype))
        kern.bilinear_bwd(g, Cn * Ho * Wo, dx, Cn * Hi * Wi, B, Cn, Hi, Wi, Ho, Wo, sh, sw, align, dx_add=g_tap)
        return (dx,) + (None,) * 6


def _f32(v: float) -> float:
    return float(torch.tensor(v, dtype=torch.float32))


def interpolate_bilinear(x, size=None, scale_factor=None, align_corners=False, tap=False):
    """F.interpolate(mode='bilinear') with PyTorch's coordinate rules (recompute_scale_factor=None).
    tap=True returns (y, x_tap): hand x_tap (not x) to x's other consumers"""
    Hi, Wi = x.shape[2:]
    if size is not None:
        Ho, Wo = size
        sfh = sfw = None
    else:
        Ho, Wo = int(math.floor(Hi * scale_factor)), int(math.floor(Wi * scale_factor))
        sfh = sfw = scale_factor
    if align_corners:
        sh = _f32((Hi - 1) / (Ho - 1)) if Ho > 1 else 0.0
        sw = _f32((Wi - 1) / (Wo - 1)) if Wo > 1 else 0.0
    else:
        sh = _f32(1.0 / sfh) if sfh else _f32(Hi / Ho)
        sw = _f32(1.0 / sfw) if sfw else _f32(Wi / Wo)
    return BilinearFn.apply(x, Ho, Wo, sh, sw, int(align_corners), tap)


class Nearest2xFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, Cn, Hi, Wi = x.shape
        y = _act((B, Cn, 2 * Hi, 2 * Wi), x)
        kern.nearest2x_fwd(x, Cn * Hi * Wi, y, 4 * Cn * Hi * Wi, B, Cn, Hi, Wi)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        B, Cn, Ho, Wo = g.shape
        Hi, Wi = Ho // 2, Wo // 2
        dx = _act((B, Cn, Hi, Wi), g)
        kern.nearest2x_bwd(g, Cn * Ho * Wo, dx, Cn * Hi * Wi, B, Cn, Hi, Wi)
        return dx


def nearest2x(x):
    return Nearest2xFn.apply(x)


class EucbFrontFn(Function):
    """blocks.py:297-321 up to the 1x1 conv — LeakyReLU(BatchNorm(DW3x3(nearest_x2(x)))) — as ONE launch per pass
    (csrc/chanloc.hip: workgroup = channel over the whole batch; the up-sampled tensor and the conv output never exist in HBM,
    the backward recomputes them from x)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, rmean, rvar, nbt, eps, slope, momentum):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        y = _act((B, Cn, 2 * H, 2 * Wd), x)
        mean, var = _empty((Cn,), x), _empty((Cn,), x)
        kern.eucb_fwd(x, w, gamma, beta, eps, slope, y, mean, var, rmean, rvar, momentum, nbt, B, Cn, H, Wd)
        ctx.save_for_backward(x, w, gamma, beta, mean, var)
        ctx.refs = (w, gamma, beta)
        ctx.cfg = (eps, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, gamma, beta, mean, var = ctx.saved_tensors
        wp, gp, bp = ctx.refs
        eps, slope = ctx.cfg
        g = _c(g)
        B, Cn, H, Wd = x.shape
        dx = torch.empty_like(x)
        dw, dg, db = grad_buf(wp), grad_buf(gp), grad_buf(bp)
        if dw is None:
            dw = _zeros(w.shape, x)
        if dg is None:
            dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
        kern.eucb_bwd(g, x, w, gamma, beta, eps, slope, mean, var, dx, dw, dg, db, B, Cn, H, Wd)
        return (dx,) + (None,) * 9


def eucb_front_supported(x, training: bool) -> bool:
    return bool(training) and x.dim() == 4 and kern.eucb_supported(x)


def eucb_front(x, w, gamma, beta, rmean, rvar, nbt, eps, slope, momentum):
    return EucbFrontFn.apply(x, w, gamma, beta, rmean, rvar, nbt, eps, slope, momentum)


class CfamMidFn(Function):
    """cfam.py:368-372 around nlb.py:141-148 — BatchNorm of the Non-local block's output conv, the residual mix (1 - w) m + w p,
    the layer-scale residual x0 + ls1 * (...) and norm2 — as ONE launch per pass (csrc/chanloc.hip, workgroup = channel over the
    batch).  -> (x1, y2)"""

    @staticmethod
    def forward(ctx, p_raw, m, x0, w, ls, bnp, bn2):
        p_raw, m, x0 = _c(p_raw), _c(m), _c(x0)
        B, Cn = x0.shape[:2]
        HW = x0.numel() // (B * Cn)
        x1, y2 = torch.empty_like(x0), torch.empty_like(x0)
        meanp, varp, mean2, var2 = (_empty((Cn,), x0) for _ in range(4))
        kern.cfam_mid_fwd(p_raw, m, x0, x1, y2, bnp.weight, bnp.bias, bnp.eps, meanp, varp, bnp.running_mean, bnp.running_var,
                          _mom(bnp), bnp.num_batches_tracked, w, ls, bn2.weight, bn2.bias, bn2.eps, mean2, var2, bn2.running_mean,
                          bn2.running_var, _mom(bn2), bn2.num_batches_tracked, B, Cn, HW)
        ctx.save_for_backward(p_raw, m, x1, w, ls, bnp.weight, bnp.bias, bn2.weight, meanp, varp, mean2, var2)
        ctx.refs = (w, ls, bnp.weight, bnp.bias, bn2.weight, bn2.bias)
        ctx.cfg = (bnp.eps, bn2.eps)
        return x1, y2

    @staticmethod
    def backward(ctx, g_x1, g_y2):
        p_raw, m, x1, w, ls, gp, bp, g2, meanp, varp, mean2, var2 = ctx.saved_tensors
        wp, lsp, gpp, bpp, g2p, b2p = ctx.refs
        epsp, eps2 = ctx.cfg
        if g_y2 is None:
            raise RuntimeError("cfam_mid: the normalised output carried no gradient")
        g_y2 = _c(g_y2)
        if g_x1 is not None:
            g_x1 = _c(g_x1) if g_x1.dtype == g_y2.dtype else _c(g_x1.to(g_y2.dtype))
        B, Cn = x1.shape[:2]
        HW = x1.numel() // (B * Cn)
        d_p, d_m, d_x0 = torch.empty_like(x1), torch.empty_like(x1), torch.empty_like(x1)
        kern.cfam_mid_bwd(g_y2, g_x1, p_raw, m, x1, d_p, d_m, d_x0, gp, bp, epsp, meanp, varp, w, ls, g2, eps2, mean2, var2,
                          _gb(gpp, x1), _gb(bpp, x1), _gb(wp, x1), _gb(lsp, x1), _gb(g2p, x1), _gb(b2p, x1), B, Cn, HW)
        return d_p, d_m, d_x0, None, None, None, None


def cfam_mid_supported(x, bnp, bn2) -> bool:
    return (bnp.training and bn2.training and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)
            and kern.chanloc_supported(x.shape[0], x.shape[2] * x.shape[3]))


def cfam_mid(p_raw, m, x0, w, ls, bnp, bn2):
    """bnp / bn2: the nn.BatchNorm2d containers (parameters, buffers, eps, momentum)"""
    return CfamMidFn.apply(p_raw, m, x0, w, ls, bnp, bn2)


class CfamFrontFn(Function):
    """cfam.py:366 over cfam.py:251-264: norm1 and the CCU gate as ONE launch per pass (csrc/chanloc.hip).
    -> (y1 = BatchNorm_1(x0): the MCA shortcut, xs = CCU(y1), x0 itself as a tap for the residual around the block)"""

    @staticmethod
    def forward(ctx, x0, bn1, ccu):
        x0 = _c(x0)
        B, Cn, H, Wd = x0.shape
        HW = H * Wd
        y1, xs = torch.empty_like(x0), torch.empty_like(x0)
        mean1, var1 = _empty((Cn,), x0), _empty((Cn,), x0)
        u, z, zn = _empty((B, Cn, 3), x0), _empty((B, Cn), x0), _empty((B, Cn), x0)
        amax = _empty((B, Cn), x0, torch.int32)
        bd = ccu.bn
        use_bn = B > 1 and not _Batch1.on
        meand, vard = (_empty((Cn,), x0), _empty((Cn,), x0)) if use_bn else (None, None)
        kern.cfam_front_fwd(x0, y1, xs, bn1.weight, bn1.bias, bn1.eps, mean1, var1, bn1.running_mean, bn1.running_var, _mom(bn1),
                            bn1.num_batches_tracked, ccu.fc1.weight, ccu.fc2.weight, bd.weight if use_bn else None,
                            bd.bias if use_bn else None, bd.eps, meand, vard, bd.running_mean if use_bn else None,
                            bd.running_var if use_bn else None, _mom(bd), bd.num_batches_tracked if use_bn else None, u, amax, z, zn,
                            B, Cn, HW)
        ctx.save_for_backward(x0, bn1.weight, bn1.bias, mean1, var1, ccu.fc1.weight, ccu.fc2.weight, bd.weight, meand, vard, u,
                              amax, z, zn)
        ctx.refs = (bn1.weight, bn1.bias, ccu.fc1.weight, ccu.fc2.weight, bd.weight, bd.bias)
        ctx.cfg = (bn1.eps, use_bn, bd.eps)
        return y1, xs, x0.view_as(x0)

    @staticmethod
    def backward(ctx, g_y1, g_xs, g_tap):
        x0, g1, b1, mean1, var1, fc1, fc2, gd, meand, vard, u, amax, z, zn = ctx.saved_tensors
        eps1, use_bn, epsd = ctx.cfg
        if g_xs is None:
            raise RuntimeError("cfam_front: the gated output carried no gradient")
        g_xs = _c(g_xs)
        fix = lambda t: None if t is None else (_c(t) if t.dtype == g_xs.dtype else _c(t.to(g_xs.dtype)))  # noqa: E731
        g_y1, g_tap = fix(g_y1), fix(g_tap)
        B, Cn = x0.shape[:2]
        HW = x0.numel() // (B * Cn)
        dx0 = torch.empty_like(x0)
        r = ctx.refs
        kern.cfam_front_bwd(g_xs, g_y1, g_tap, x0, dx0, g1, b1, eps1, mean1, var1, fc1, fc2, gd if use_bn else None, epsd, meand,
                            vard, u, amax, z, zn, _gb(r[0], x0), _gb(r[1], x0), _gb(r[2], x0), _gb(r[3], x0),
                            _gb(r[4], x0) if use_bn else None, _gb(r[5], x0) if use_bn else None, B, Cn, HW)
        return dx0, None, None


def cfam_front_supported(x, bn1, ccu) -> bool:
    return (bn1.training and ccu.bn.training and x.dim() == 4 and x.shape[0] <= 256
            and kern.chanloc_supported(x.shape[0], x.shape[2] * x.shape[3]))


def cfam_front(x0, bn1, ccu):
    """bn1: the block's norm1 (nn.BatchNorm2d), ccu: its CCU module (fc1, fc2, bn) -> (y1, xs, x0_tap)"""
    return CfamFrontFn.apply(x0, bn1, ccu)


_POOL_R: dict = {}


def _bil_matrix(n_in: int, n_out: int, scale: float, align: bool) -> Tensor:
    """[n_out, n_in] fp32 matrix of one bilinear resampling along an axis, by the kernels' coordinate rule (resample.hip
    bil_coord: align: src = scale * dst; else src = max(scale * (dst + 0.5) - 0.5, 0); fp32 arithmetic)"""
    R = torch.zeros(n_out, n_in, dtype=torch.float32)
    sc = torch.tensor(scale, dtype=torch.float32)
    for d in range(n_out):
        dst = torch.tensor(float(d), dtype=torch.float32)
        src = sc * dst if align else torch.clamp(sc * (dst + 0.5) - 0.5, min=0.0)
        i0 = min(int(src.item()), n_in - 1)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        l1 = min(float((src - i0).item()), 1.0)
        R[d, i0] += 1.0 - l1
        R[d, i1] += l1
    return R


def _pool_matrices(H: int, Wd: int, device):
    """(RH [H, 7], RW [W, 7]): cfam.py:217 (UpsamplingBilinear2d x7, align_corners=True) followed by cfam.py:232 (interpolate to
    (H, W), align_corners=False, skipped when 49 == H) composed into one linear map per axis"""
    key = (H, Wd, str(device))
    m = _POOL_R.get(key)
    if m is None:
        def one(n):
            r1 = _bil_matrix(7, 49, _f32(6.0 / 48.0), True).double()
            r = r1 if n == 49 else _bil_matrix(49, n, _f32(49.0 / n), False).double() @ r1
            return r.float().contiguous().to(device)
        m = _POOL_R[key] = (one(H), one(Wd))
    return m


class PoolBranchFn(Function):
    """cfam.py:212-218,231-232: AdaptiveAvgPool(7) -> 1x1 conv -> BatchNorm -> LeakyReLU(0.01) -> x7 bilinear (align) -> bilinear to
    (H, W): two launches per pass (csrc/chanloc.hip pool_mix_* / pool_up_*) instead of six."""

    @staticmethod
    def forward(ctx, x, wc, bn, H, Wd):
        x = _c(x)
        B, P = x.shape[:2]
        RH, RW = _pool_matrices(H, Wd, x.device)
        y = torch.empty_like(x)
        pooled, t = _empty((B, P, 49), x), _empty((B, P, 49), x)
        mean, var = _empty((P,), x), _empty((P,), x)
        kern.pool_branch_fwd(x, P * H * Wd, wc, bn.weight, bn.bias, bn.eps, 0.01, RH, RW, y, P * H * Wd, pooled, t, mean, var,
                             bn.running_mean, bn.running_var, _mom(bn), bn.num_batches_tracked, B, P, H, Wd)
        ctx.save_for_backward(wc, bn.weight, bn.bias, RH, RW, pooled, t, mean, var)
        ctx.refs = (wc, bn.weight, bn.bias)
        ctx.cfg = (bn.eps, B, P, H, Wd)
        return y

    @staticmethod
    def backward(ctx, g):
        wc, gamma, beta, RH, RW, pooled, t, mean, var = ctx.saved_tensors
        eps, B, P, H, Wd = ctx.cfg
        g = _c(g)
        dx = torch.empty_like(g)
        dt = _empty((B, P, 49), g)
        r = ctx.refs
        kern.pool_branch_bwd(g, P * H * Wd, wc, gamma, beta, eps, 0.01, RH, RW, pooled, t, mean, var, dt, dx, P * H * Wd,
                             _gb(r[0], g), _gb(r[1], g), _gb(r[2], g), B, P, H, Wd)
        return dx, None, None, None, None


class JoinBnPoolFn(Function):
    """Tail of MultiOrderDWConv's branches without the concat (cfam.py:233-240): the (merged) pointwise BatchNorm + ReLU of the three
    dilated branches writes channels [0, 3g) of ONE [B, C, H, W] tensor, the pooled branch (PoolBranchFn's kernels) channels
    [3g, C); backwards both read their slice of the joint gradient in place (batch strides) — no cat / split launches."""

    @staticmethod
    def forward(ctx, v_raw, rest, gamma, beta, rmean, rvar, nbt, eps, momentum, wc, pbn):
        v_raw, rest = _c(v_raw), _c(rest)
        B, G3, H, Wd = v_raw.shape
        P = rest.shape[1]
        Cn, HW = G3 + P, H * Wd
        joint = _act((B, Cn, H, Wd), v_raw)
        mean, var = _empty((G3,), v_raw), _empty((G3,), v_raw)
        ws = _empty((2 * G3 * 256,), v_raw)  # CENET_BN_WS_FLOATS(C)
        kern.bn_train_fwd(v_raw, G3 * HW, joint, Cn * HW, ws, mean, var, rmean, rvar, momentum, nbt, eps, gamma, beta, "relu", 0.0, B,
                          G3, HW)
        RH, RW = _pool_matrices(H, Wd, v_raw.device)
        pooled, t = _empty((B, P, 49), v_raw), _empty((B, P, 49), v_raw)
        pmean, pvar = _empty((P,), v_raw), _empty((P,), v_raw)
        kern.pool_branch_fwd(rest, P * HW, wc, pbn.weight, pbn.bias, pbn.eps, 0.01, RH, RW, kern.Ptr(joint, G3 * HW), Cn * HW, pooled,
                             t, pmean, pvar, pbn.running_mean, pbn.running_var, _mom(pbn), pbn.num_batches_tracked, B, P, H, Wd)
        ctx.save_for_backward(v_raw, gamma, beta, mean, var, wc, pbn.weight, pbn.bias, RH, RW, pooled, t, pmean, pvar)
        ctx.refs = (gamma, beta, wc, pbn.weight, pbn.bias)
        ctx.cfg = (eps, pbn.eps, P)
        return joint

    @staticmethod
    def backward(ctx, g):
        v_raw, gamma, beta, mean, var, wc, pg, pb, RH, RW, pooled, t, pmean, pvar = ctx.saved_tensors
        eps, peps, P = ctx.cfg
        g = _c(g)
        B, G3, H, Wd = v_raw.shape
        Cn, HW = G3 + P, H * Wd
        r = ctx.refs
        dv = torch.empty_like(v_raw)
        ws = _empty((2 * G3 * 256,), v_raw)
        kern.bn_bwd(g, Cn * HW, v_raw, G3 * HW, dv, G3 * HW, mean, var, eps, gamma, beta, "relu", 0.0, B, G3, HW, ws, _gb(r[0], g),
                    _gb(r[1], g))
        drest = torch.empty((B, P, H, Wd), device=g.device, dtype=g.dtype)
        dt = _empty((B, P, 49), g)
        kern.pool_branch_bwd(kern.Ptr(g, G3 * HW), Cn * HW, wc, pg, pb, peps, 0.01, RH, RW, pooled, t, pmean, pvar, dt, drest, P * HW,
                             _gb(r[2], g), _gb(r[3], g), _gb(r[4], g), B, P, H, Wd)
        return (dv, drest) + (None,) * 9


def join_bn_pool(v_raw, rest, gamma, beta, rmean, rvar, nbt, eps, momentum, wc, pbn):
    """-> [B, 3g + p, H, W]: ReLU(BatchNorm(v_raw)) | pooled_branch(rest)"""
    return JoinBnPoolFn.apply(v_raw, rest, gamma, beta, rmean, rvar, nbt, eps, momentum, wc, pbn)


def pool_branch_supported(x, bn) -> bool:
    return bool(bn.training) and x.dim() == 4 and kern.pool_branch_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3])


def pool_branch(x, wc, bn):
    """x [B, p, H, W] (the pooled branch's channel slice), wc [p, p, 1, 1], bn: its nn.BatchNorm2d"""
    return PoolBranchFn.apply(x, wc, bn, x.shape[2], x.shape[3])


class AdaptiveAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x = _c(x)
        B, Cn, Hi, Wi = x.shape
        y = _act((B, Cn, Ho, Wo), x)
        kern.avgpool_fwd(x, Cn * Hi * Wi, y, Cn * Ho * Wo, B, Cn, Hi, Wi, Ho, Wo)
        ctx.cfg = (Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, g):
        Hi, Wi, Ho, Wo = ctx.cfg
        g = _c(g)
        B, Cn = g.shape[:2]
        dx = _act((B, Cn, Hi, Wi), g)
        kern.avgpool_bwd(g, Cn * Ho * Wo, dx, Cn * Hi * Wi, B, Cn, Hi, Wi, Ho, Wo)
        return dx, None, None


def adaptive_avgpool(x, Ho, Wo):
    return AdaptiveAvgPoolFn.apply(x, Ho, Wo)


class MaxPool2ScaleFn(Function):
    """out.py:43,70: w[c] * MaxPool2d(2,2)(x)."""

    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        B, Cn, Hi, Wi = x.shape
        y = _act((B, Cn, Hi // 2, Wi // 2), x)
        kern.maxpool2_fwd(x, y, Cn * (Hi // 2) * (Wi // 2), w, B, Cn, Hi, Wi)
        ctx.save_for_backward(x, w)
        ctx.refs = (w,)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _c(g)
        B, Cn, Hi, Wi = x.shape
        dx = torch.empty_like(x)
        kern.maxpool2_bwd(x, g, Cn * (Hi // 2) * (Wi // 2), dx, w, grad_buf(ctx.refs[0]), B, Cn, Hi, Wi)
        return dx, None


def maxpool2_scale(x, w):
    return MaxPool2ScaleFn.apply(x, w)


class ResTailPoolFn(Function):
    """w * MaxPool2d(2,2)(LeakyReLU(BN2(x2) + BN3(x3))) — the tail of the head's image branch (out.py:60,70 over unet.py:201-214)
    on bf16 maps in training mode: batch statistics by the BatchNorm statistics kernels, then ONE pass (csrc/res_tail.hip); the
    backward recomputes the activation from x2 / x3 in its two passes."""

    @staticmethod
    def forward(ctx, x2, x3, w, slope, g2, b2, rm2, rv2, nbt2, eps2, mom2, g3, b3, rm3, rv3, nbt3, eps3, mom3):
        x2, x3 = _c(x2), _c(x3)
        B, Cn, H, Wd = x2.shape
        HW = H * Wd
        st = []
        for x, rm, rv, nbt, mom in ((x2, rm2, rv2, nbt2, mom2), (x3, rm3, rv3, nbt3, mom3)):
            mean, var = _empty((Cn,), x), _empty((Cn,), x)
            ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
            kern.bn_stats(x, Cn * HW, B, Cn, HW, ws, mean, var, rm, rv, mom, nbt)
            st += [mean, var]
        wv = _c(w.reshape(-1))
        out = _act((B, Cn, H // 2, Wd // 2), x2)
        kern.res_tail_fwd(x2, x3, st[0], st[1], g2, b2, eps2, st[2], st[3], g3, b3, eps3, wv, slope, out, B, Cn, H, Wd)
        ctx.save_for_backward(x2, x3, wv, g2, b2, g3, b3, *st)
        ctx.refs = (w, g2, b2, g3, b3)
        ctx.cfg = (slope, eps2, eps3)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, x3, wv, g2, b2, g3, b3, m2, v2, m3, v3 = ctx.saved_tensors
        wp, g2p, b2p, g3p, b3p = ctx.refs
        slope, eps2, eps3 = ctx.cfg
        g = _c(g)
        B, Cn, H, Wd = x2.shape
        dx2, dx3 = torch.empty_like(x2), torch.empty_like(x3)
        dw = grad_buf(wp)
        kern.res_tail_bwd(g, x2, x3, m2, v2, g2, b2, eps2, m3, v3, g3, b3, eps3, wv, slope, dx2, dx3, grad_buf(g2p), grad_buf(b2p),
                          grad_buf(g3p), grad_buf(b3p), dw.view(-1) if dw is not None else None, B, Cn, H, Wd)
        return (dx2, dx3) + (None,) * 16


def res_tail_pool(x2, bn2, x3, bn3, w, slope):
    """bn2 / bn3: nn.BatchNorm2d modules in training mode (their running statistics are updated)"""
    return ResTailPoolFn.apply(x2, x3, w, slope, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked,
                               bn2.eps, bn_momentum(bn2), bn3.weight, bn3.bias, bn3.running_mean,
                               bn3.running_var, bn3.num_batches_tracked, bn3.eps, bn_momentum(bn3))


class ResTailImgPoolFn(Function):
    """ResTailPoolFn for a ONE-CHANNEL network input: the shortcut x3 = conv3(img) = w3[c] * img (unet.py conv3, 1x1) is not
    materialised — BN3(x3) is an affine map of the image, with coefficients from the image's batch statistics.  No shortcut conv, no
    statistics pass over it, no dx3, no weight-gradient kernel for it: its weight only reaches the output through eps (BatchNorm is
    invariant to the scale of its input), and that gradient comes out of the sums the backward computes anyway."""
    _dummy: dict = {}  # per device: (running mean, running var, counter) the image's statistics call writes nowhere useful

    @staticmethod
    def forward(ctx, x2, img, w3, w, slope, g2, b2, rm2, rv2, nbt2, eps2, mom2, g3, b3, rm3, rv3, nbt3, eps3, mom3):
        x2, img = _c(x2), _c(img)
        B, Cn, H, Wd = x2.shape
        HW = H * Wd
        mean2, var2 = _empty((Cn,), x2), _empty((Cn,), x2)
        kern.bn_stats(x2, Cn * HW, B, Cn, HW, _empty((2 * Cn * 256,), x2), mean2, var2, rm2, rv2, mom2, nbt2)
        key = (x2.device.type, x2.device.index)
        dm = ResTailImgPoolFn._dummy.get(key)
        if dm is None:
            dm = ResTailImgPoolFn._dummy[key] = (torch.zeros(1, device=x2.device), torch.ones(1, device=x2.device),
                                                 torch.zeros(1, device=x2.device, dtype=torch.long))
        imean, ivar = _empty((1,), x2), _empty((1,), x2)
        kern.bn_stats(img, HW, B, 1, HW, _empty((2 * 256,), x2), imean, ivar, dm[0], dm[1], 0.0, dm[2])
        wv, w3v = _c(w.reshape(-1)), _c(w3.reshape(-1))
        out = _act((B, Cn, H // 2, Wd // 2), x2)
        kern.res_tail_img_fwd(x2, img, mean2, var2, g2, b2, eps2, imean, ivar, w3v, g3, b3, eps3, rm3, rv3, nbt3, mom3, wv, slope, out,
                              B, Cn, H, Wd)
        ctx.save_for_backward(x2, img, wv, w3v, g2, b2, g3, b3, mean2, var2, imean, ivar)
        ctx.refs = (w, w3, g2, b2, g3, b3)
        ctx.cfg = (slope, eps2, eps3)
        return out

    @staticmethod
    def backward(ctx, g):
        x2, img, wv, w3v, g2, b2, g3, b3, mean2, var2, imean, ivar = ctx.saved_tensors
        wp, w3p, g2p, b2p, g3p, b3p = ctx.refs
        slope, eps2, eps3 = ctx.cfg
        g = _c(g)
        B, Cn, H, Wd = x2.shape
        dx2 = torch.empty_like(x2)
        dw, dw3 = grad_buf(wp), grad_buf(w3p)
        kern.res_tail_img_bwd(g, x2, img, mean2, var2, g2, b2, eps2, imean, ivar, w3v, g3, b3, eps3, wv, slope, dx2, grad_buf(g2p),
                              grad_buf(b2p), grad_buf(g3p), grad_buf(b3p), dw3.view(-1) if dw3 is not None else None,
                              dw.view(-1) if dw is not None else None, B, Cn, H, Wd)
        return (dx2,) + (None,) * 18


def res_tail_img_pool(x2, bn2, img, w3, bn3, w, slope):
    return ResTailImgPoolFn.apply(x2, img, w3, w, slope, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var,
                                  bn2.num_batches_tracked, bn2.eps, bn_momentum(bn2), bn3.weight,
                                  bn3.bias, bn3.running_mean, bn3.running_var, bn3.num_batches_tracked, bn3.eps,
                                  bn_momentum(bn3))


def res_tail_img_pool_supported(x2, img, w3, bn2, bn3, w) -> bool:
    """conv3 is a bias-free 1x1 conv of a one-channel bf16 image that needs no gradient"""
    return bool(bn2.training and bn3.training and _bf(x2) and _bf(img) and img.dim() == 4 and img.shape[1] == 1
                and not img.requires_grad and tuple(w3.shape[1:]) == (1, 1, 1) and w3.shape[0] == x2.shape[1]
                and img.shape[0] == x2.shape[0] and tuple(img.shape[2:]) == tuple(x2.shape[2:]) and w.numel() == x2.shape[1]
                and x2.data_ptr() % 16 == 0 and img.data_ptr() % 16 == 0
                and kern._lib.lib().cenet_res_tail_supported(int(x2.shape[2]), int(x2.shape[3]))
                and os.environ.get("CENET_RES_TAIL_FUSED", "1") not in ("0", "x3"))


def res_tail_pool_supported(x2, x3, bn2, bn3, w) -> bool:
    return bool(bn2.training and bn3.training and x2.is_cuda == x3.is_cuda and kern.res_tail_supported(x2, x3)
                and w.numel() == x2.shape[1] and os.environ.get("CENET_RES_TAIL_FUSED", "1") != "0")


__all__ = [n for n in dir() if not n.startswith("__")]
