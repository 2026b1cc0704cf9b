"""cenet_amd.ops.attention — attention operators: spatial-reduction attention, Non-local attention, differential attention heads + combine.
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


# =====================================================================================================
# attention
# =====================================================================================================
class _AttnDesc:
    """strides/dims of one attention problem; tensors are addressed in place."""

    def __init__(self, B, H, Nq, Nk, D, Dv, scale, vdiv, qs, ks, vs, os_, qoff=0, koff=0, voff=0):
        self.B, self.H, self.Nq, self.Nk, self.D, self.Dv, self.scale, self.vdiv = B, H, Nq, Nk, D, Dv, scale, vdiv
        self.qs, self.ks, self.vs, self.os = qs, ks, vs, os_  # each (sb, sh, si, sd)
        self.qoff, self.koff, self.voff = qoff, koff, voff   # element offsets into q/k/v storage
        self.finite = False  # fp32 flash forward: q scaled first + torch.nan_to_num of the scores (multihead_diffattn.py:95,106)

    def fill(self, a: "kern.AttnT", q, k, v, o, lse):
        e = kern.esz(q)
        a.q, a.k, a.v = q.data_ptr() + e * self.qoff, k.data_ptr() + e * self.koff, v.data_ptr() + e * self.voff
        a.o, a.lse = o.data_ptr(), lse.data_ptr() if lse is not None else None
        a.qsb, a.qsh, a.qsi, a.qsd = self.qs
        a.ksb, a.ksh, a.ksi, a.ksd = self.ks
        a.vsb, a.vsh, a.vsi, a.vsd = self.vs
        a.osb, a.osh, a.osi, a.osd = self.os
        a.B, a.H, a.Nq, a.Nk, a.D, a.Dv, a.v_head_div = self.B, self.H, self.Nq, self.Nk, self.D, self.Dv, self.vdiv
        a.scale = self.scale
        a.finite_scores = int(self.finite)


def _attn_forward(d: _AttnDesc, q, k, v, o):
    """returns the tensors to save for backward: ('flash', lse) or ('mat', P)."""
    kern._chk(q, k, v, o)
    bf = _bf(q)
    if kern.flashb_supported(d.D, d.Dv) if bf else kern.flash_supported(d.D, d.Dv):
        lse = _empty((d.B, d.H, d.Nq), q)
        a = kern.AttnT()
        d.fill(a, q, k, v, o, lse)
        kern.flash_fwd(a, bf)
        return "flash", lse
    # materialised path (large head dims, small N): S = scale QK^T ; P = softmax ; O = P V.  The scores stay fp32 in both
    # modes (bf16 operands: the GEMM adds atomically into a zero-filled fp32 S); P has the operand type.
    BH = d.B * d.H
    S = _empty((BH, d.Nq, d.Nk), q)  # (bf16 operands: the GEMM STORES its fp32 accumulators, atomic=2 — no zero fill)
    kern.gemm(kern.mat_plain(q, d.qs[2], d.qs[3], sb=d.qs[0], sb2=d.qs[1], kfast=int(d.qs[3] == 1), offset=d.qoff),
              kern.mat_plain(k, d.ks[3], d.ks[2], sb=d.ks[0], sb2=d.ks[1], kfast=int(d.ks[3] == 1), offset=d.koff),
              S, d.Nq, d.Nk, d.D, scr=d.Nk, scc=1, scb=d.H * d.Nq * d.Nk, scb2=d.Nq * d.Nk, nbatch=BH, nb_inner=d.H,
              alpha=d.scale, atomic=2 if bf else False)
    P = _act(S.shape, q)
    kern.softmax_rows_fwd(S, P, BH * d.Nq, d.Nk)
    vh = d.vdiv
    kern.gemm(kern.mat_plain(P, d.Nk, 1, sb=d.H * d.Nq * d.Nk, sb2=d.Nq * d.Nk, kfast=1),
              _vmat(d, v, vh), o, d.Nq, d.Dv, d.Nk, scr=d.os[2], scc=d.os[3], scb=d.os[0], scb2=d.os[1], nbatch=BH,
              nb_inner=d.H)
    return "mat", P


def _vmat(d: _AttnDesc, v, vh):
    # B operand V[j, dv] for batch (b, h): head h // vh.  nb_inner=H indexes h, so use stride vs[1]/vh when vh divides
    # evenly is not expressible -> when vh > 1 the caller loops over the two heads (see _attn_* below).
    assert vh == 1
    return kern.mat_plain(v, d.vs[2], d.vs[3], sb=d.vs[0], sb2=d.vs[1], kfast=int(d.vs[2] == 1), offset=d.voff)


def _attn_backward(d: _AttnDesc, kind, saved, q, k, v, o, g, dq, dk, dv, dkv_zeroed=False):
    """dq/dk/dv are written in the layouts of q/k/v (offsets included); dv must be zero-filled when vdiv > 1.
    dkv_zeroed: dk and dv are zero-filled (lets the kernels split the query range when there are few keys)."""
    bf = _bf(q)
    if kind == "flash":
        a = kern.AttnT()
        d.fill(a, q, k, v, o, saved)
        a.dkv_zeroed = int(dkv_zeroed)
        delta = _empty((d.B, d.H, d.Nq), q)
        a.dout = g.data_ptr()
        a.dq = dq.data_ptr() + kern.esz(dq) * d.qoff
        a.dk, a.dv = dk.data_ptr() + kern.esz(dk) * d.koff, dv.data_ptr() + kern.esz(dv) * d.voff
        a.dkv_f32 = int(bf and dk.dtype == torch.float32)
        assert dk.dtype == dv.dtype
        a.delta = delta.data_ptr()
        kern.flash_bwd(a, bf)
        return
    P = saved
    BH = d.B * d.H
    HN = d.H * d.Nq * d.Nk
    NN = d.Nq * d.Nk
    dP = torch.empty(P.shape, device=P.device, dtype=torch.float32)  # fp32 in both modes (see _attn_forward)
    kern.gemm(kern.mat_plain(g, d.os[2], d.os[3], sb=d.os[0], sb2=d.os[1], kfast=int(d.os[3] == 1)),
              kern.mat_plain(v, d.vs[3], d.vs[2], sb=d.vs[0], sb2=d.vs[1], kfast=int(d.vs[3] == 1), offset=d.voff),
              dP, d.Nq, d.Nk, d.Dv, scr=d.Nk, scc=1, scb=HN, scb2=NN, nbatch=BH, nb_inner=d.H, atomic=2 if bf else False)
    dS = torch.empty_like(P)
    kern.softmax_rows_bwd(P, dP, dS, BH * d.Nq, d.Nk)
    kern.gemm(kern.mat_plain(dS, d.Nk, 1, sb=HN, sb2=NN, kfast=1),
              kern.mat_plain(k, d.ks[2], d.ks[3], sb=d.ks[0], sb2=d.ks[1], kfast=int(d.ks[2] == 1), offset=d.koff),
              dq, d.Nq, d.D, d.Nk, scr=d.qs[2], scc=d.qs[3], scb=d.qs[0], scb2=d.qs[1], nbatch=BH, nb_inner=d.H,
              alpha=d.scale, c_offset=d.qoff)
    kern.gemm(kern.mat_plain(dS, 1, d.Nk, sb=HN, sb2=NN, kfast=0),
              kern.mat_plain(q, d.qs[2], d.qs[3], sb=d.qs[0], sb2=d.qs[1], kfast=int(d.qs[2] == 1), offset=d.qoff),
              dk, d.Nk, d.D, d.Nq, scr=d.ks[2], scc=d.ks[3], scb=d.ks[0], scb2=d.ks[1], nbatch=BH, nb_inner=d.H,
              alpha=d.scale, c_offset=d.koff)
    kern.gemm(kern.mat_plain(P, 1, d.Nk, sb=HN, sb2=NN, kfast=0),
              kern.mat_plain(g, d.os[2], d.os[3], sb=d.os[0], sb2=d.os[1], kfast=int(d.os[2] == 1)),
              dv, d.Nk, d.Dv, d.Nq, scr=d.vs[2], scc=d.vs[3], scb=d.vs[0], scb2=d.vs[1], nbatch=BH, nb_inner=d.H,
              c_offset=d.voff)


class SRAttentionFn(Function):
    """pvtv2.py:88-109 core: q [B,N,C] (heads x hd), kv [B,Nk,2C] = [k | v] -> out [B,N,C]."""

    @staticmethod
    def forward(ctx, q, kv, heads):
        q, kv = _c(q), _c(kv)
        B, N, Cn = q.shape
        Nk = kv.shape[1]
        hd = Cn // heads
        d = _AttnDesc(B, heads, N, Nk, hd, hd, hd ** -0.5, 1, (N * Cn, hd, Cn, 1), (Nk * 2 * Cn, hd, 2 * Cn, 1),
                      (Nk * 2 * Cn, hd, 2 * Cn, 1), (N * Cn, hd, Cn, 1), voff=Cn)
        o = torch.empty_like(q)
        if _bf(q) and kern.sra_attn_bwd_supported(hd, Nk):
            # bf16, 64-dim heads, <= 64 keys: keys / values resident, no key loop (attn_diff.hip, sra_fwd_kernel); the lse it
            # leaves is the tiled kernels' (kind "flash": the backward below does not care which forward ran)
            saved = _empty((B, heads, N), q)
            kern.sra_attn_fwd(q, kv, o, saved, B, heads, N, Nk, d.scale)
            kind = "flash"
        else:
            kind, saved = _attn_forward(d, q, kv, kv, o)
        ctx.save_for_backward(q, kv, o, saved)
        ctx.d, ctx.kind = d, kind
        return o

    @staticmethod
    def backward(ctx, g):
        q, kv, o, saved = ctx.saved_tensors
        g = _c(g)
        d = ctx.d
        if ctx.kind == "flash" and _bf(q) and kern.sra_attn_bwd_supported(d.D, d.Nk) and d.D == d.Dv:
            # bf16, 64-dim heads, <= 64 keys: one kernel with the key / value set resident (attn_diff.hip, sra_bwd_kernel)
            dq = torch.empty_like(q)
            if kern.sra_attn_bwd_direct_supported(d.B, d.H, d.Nq, d.Nk):  # one workgroup per (batch, head): bf16 dK/dV directly
                dkv = torch.empty_like(kv)
                kern.sra_attn_bwd_direct(q, kv, o, g, saved, dq, dkv, d.B, d.H, d.Nq, d.Nk, d.scale)
                return dq, dkv, None
            dkv = _ZeroWs.take(kv.shape, kv)  # (zero at rest: no fill launch; cast_clear leaves it zero again)
            kern.sra_attn_bwd(q, kv, o, g, saved, dq, dkv, d.B, d.H, d.Nq, d.Nk, d.scale)
            return dq, _ZeroWs.give_back_as(dkv, kv), None
        if (ctx.kind == "flash" and _bf(q) and kern.sra_attn_bwd_blocks_supported(d.D, d.Nk) and d.D == d.Dv
                and all(t.data_ptr() % 16 == 0 for t in (q, kv, o, g))):
            # bf16, 64-dim heads, 65 .. 256 keys (512x512 inputs): dQ with all keys resident + dK / dV per 64-key block
            dq = torch.empty_like(q)
            dkv = _zeros(kv.shape, kv)
            kern.sra_attn_bwd(q, kv, o, g, saved, dq, dkv, d.B, d.H, d.Nq, d.Nk, d.scale)
            return dq, kern.cast(dkv, kv.dtype), None
        few_keys = ctx.d.Nk <= 128 and ctx.d.Nq >= 1024  # spatial-reduction attention: 49 keys under 784..3136 queries
        # (the query range is then sliced over workgroups and dK / dV are added atomically: fp32 accumulator)
        dq, dkv = torch.empty_like(q), (_zeros(kv.shape, kv) if few_keys else torch.empty_like(kv))
        _attn_backward(ctx.d, ctx.kind, saved, q, kv, kv, o, g, dq, dkv, dkv, dkv_zeroed=few_keys)
        return dq, kern.cast(dkv, kv.dtype), None


class NonlocalAttnFn(Function):
    """nlb.py:117-138: theta, phi, g [B,C,N] channel-major -> y[b,c,i] = sum_j softmax_j(theta_i.phi_j / sqrt(C)) g[c,j]."""

    @staticmethod
    def forward(ctx, theta, phi, gx):
        theta, phi, gx = _c(theta), _c(phi), _c(gx)
        B, Cn = theta.shape[:2]
        N = theta.numel() // (B * Cn)
        ctx.tok64 = _bf(theta) and Cn in (64, 128) and N >= 256
        if ctx.tok64:
            # bf16, C = 64 / 128 (the 56x56 / 28x28 levels): token-major copies through the single-softmax form of the pair kernels
            # (attn_diff.hip); three [64, N] -> [N, 64] transposes in, one out (~10 us each against ~2 ms saved)
            qt, kt, vt = (torch.empty((B, N, Cn), device=theta.device, dtype=theta.dtype) for _ in range(3))
            for src, dst in ((theta, qt), (phi, kt), (gx, vt)):
                kern.transpose(src, Cn * N, dst, Cn * N, B, Cn, N)
            U = torch.empty((B, 1, N, Cn), device=theta.device, dtype=theta.dtype)
            lse = torch.empty((B, 1, N), device=theta.device, dtype=torch.float32)
            a = kern.DiffAttnT()
            a.q, a.k, a.v, a.U, a.lse = qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), U.data_ptr(), lse.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale = B, 1, N, Cn, Cn ** -0.5
            kern.attn64(a, backward=False)
            o = torch.empty_like(theta)
            kern.transpose(U, N * Cn, o, Cn * N, B, N, Cn)
            ctx.save_for_backward(qt, kt, vt, U, lse)
            ctx.shape = theta.shape
            return o
        st = (Cn * N, 0, 1, N)
        d = _AttnDesc(B, 1, N, N, Cn, Cn, Cn ** -0.5, 1, st, st, st, st)
        o = torch.empty_like(theta)
        kind, saved = _attn_forward(d, theta, phi, gx, o)
        ctx.save_for_backward(theta, phi, gx, o, saved)
        ctx.d, ctx.kind = d, kind
        return o

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        if ctx.tok64:
            qt, kt, vt, U, lse = ctx.saved_tensors
            B, N, Cn = qt.shape
            gt = torch.empty_like(U)
            kern.transpose(g, Cn * N, gt, N * Cn, B, Cn, N)
            dq, dk, dv = torch.empty_like(qt), torch.empty_like(kt), torch.empty_like(vt)
            ws = torch.empty(kern.attn64_ws_bytes(B, 1, N), device=g.device, dtype=torch.uint8)
            a = kern.DiffAttnT()
            a.q, a.k, a.v, a.U, a.lse, a.dU = qt.data_ptr(), kt.data_ptr(), vt.data_ptr(), U.data_ptr(), lse.data_ptr(), gt.data_ptr()
            a.dq, a.dk, a.dv, a.ws = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale = B, 1, N, Cn, Cn ** -0.5
            kern.attn64(a, backward=True)
            outs = []
            for src in (dq, dk, dv):
                dst = torch.empty(ctx.shape, device=g.device, dtype=g.dtype)
                kern.transpose(src, N * Cn, dst, Cn * N, B, N, Cn)
                outs.append(dst)
            return tuple(outs)
        theta, phi, gx, o, saved = ctx.saved_tensors
        dt, dp, dg = torch.empty_like(theta), torch.empty_like(phi), torch.empty_like(gx)
        _attn_backward(ctx.d, ctx.kind, saved, theta, phi, gx, o, g, dt, dp, dg)
        return dt, dp, dg


class DiffAttnHeadsFn(Function):
    """multihead_diffattn.py:83-109: q,k [B,N,2H,hd], v [B,N,H,2hd] -> U [B,2H,N,2hd], U[2h+s] = softmax(q_{2h+s} k^T) v_h."""

    @staticmethod
    def forward(ctx, q, k, v, H):
        q, k, v = _c(q), _c(k), _c(v)
        B, N, E = q.shape
        hd = E // H // 2
        dv = 2 * hd
        U = _act((B, 2 * H, N, dv), q)
        # (the pair kernels read 16-byte chunks: operands that are odd-element views of a larger buffer take the tiled path)
        ctx.pairs = (_bf(q) and kern.diffattn_heads_supported(hd, N)
                     and all(t.data_ptr() % 16 == 0 for t in (q, k, v)))
        if ctx.pairs:
            # bf16 tensors: the pair kernels of attn_diff.hip (both softmax heads of a value head per wave, no atomics)
            lse = _empty((B, 2 * H, N), q)
            a = kern.DiffAttnT()
            a.q, a.k, a.v, a.U, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), U.data_ptr(), lse.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale = B, H, N, hd, hd ** -0.5
            kern.diffattn_heads(a, backward=False)
            ctx.save_for_backward(q, k, v, U, lse)
            ctx.H = H
            return U
        if kern.flashb_supported(hd, dv) if _bf(q) else kern.flash_supported(hd, dv):
            d = _AttnDesc(B, 2 * H, N, N, hd, dv, hd ** -0.5, 2, (N * E, hd, E, 1), (N * E, hd, E, 1), (N * E, dv, E, 1),
                          (2 * H * N * dv, N * dv, dv, 1))
            d.finite = not _bf(q)  # parity mode reproduces multihead_diffattn.py:106; the bf16 kernels propagate non-finite scores
            kind, saved = _attn_forward(d, q, k, v, U)
            ctx.descs = [d]
            saved_list = [saved]
        else:
            # materialised path: one problem per softmax branch s in {0,1}; both read value head h
            ctx.descs, saved_list = [], []
            for s in (0, 1):
                d = _AttnDesc(B, H, N, N, hd, dv, hd ** -0.5, 1, (N * E, 2 * hd, E, 1), (N * E, 2 * hd, E, 1),
                              (N * E, dv, E, 1), (2 * H * N * dv, 2 * N * dv, dv, 1), qoff=s * hd, koff=s * hd)
                Us = U  # written through an offset view
                kind, saved = _attn_forward(d, q, k, v, _OffsetView(Us, s * N * dv))
                ctx.descs.append(d)
                saved_list.append(saved)
        ctx.kind = kind
        ctx.save_for_backward(q, k, v, U, *saved_list)
        ctx.H = H
        return U

    @staticmethod
    def backward(ctx, g):
        q, k, v, U = ctx.saved_tensors[:4]
        saved_list = ctx.saved_tensors[4:]
        g = _c(g)
        B, N, E = q.shape
        hd = E // ctx.H // 2
        dv = 2 * hd
        if ctx.pairs:
            H = ctx.H
            lse = saved_list[0]
            # one buffer for the three gradients: MultiLinearFn (the batched q / k / v projection) then runs its data gradient
            # as ONE GEMM with three K-batches
            dqkv = torch.empty((3,) + tuple(q.shape), device=q.device, dtype=q.dtype)
            dq, dk, dvv = dqkv[0], dqkv[1], dqkv[2]
            ws = torch.empty(kern.diffattn_heads_ws_bytes(B, H, N), device=q.device, dtype=torch.uint8)
            a = kern.DiffAttnT()
            a.q, a.k, a.v, a.U, a.lse, a.dU = q.data_ptr(), k.data_ptr(), v.data_ptr(), U.data_ptr(), lse.data_ptr(), g.data_ptr()
            a.dq, a.dk, a.dv, a.ws = dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), ws.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale = B, H, N, hd, hd ** -0.5
            kern.diffattn_heads(a, backward=True)
            return dq, dk, dvv, None
        dq = torch.empty_like(q)
        if len(ctx.descs) == 1:
            # the two softmax heads of a pair ADD into the gradient of their shared value head: fp32 accumulators (dk rides
            # along: one flag covers both)
            dvv = _zeros(v.shape, v)
            dk = _empty(k.shape, k) if _bf(k) else torch.empty_like(k)
            _attn_backward(ctx.descs[0], ctx.kind, saved_list[0], q, k, v, U, g, dq, dk, dvv)
            dk, dvv = kern.cast(dk, k.dtype), kern.cast(dvv, v.dtype)
        else:
            dk = torch.empty_like(k)
            dvv = None
            for s, (d, saved) in enumerate(zip(ctx.descs, saved_list)):
                tmp = torch.empty_like(v)
                _attn_backward(d, ctx.kind, saved, q, k, v, _OffsetView(U, s * N * dv), _OffsetView(g, s * N * dv), dq, dk, tmp)
                if dvv is None:
                    dvv = tmp
                else:
                    kern.copy_batched(tmp, 0, dvv, 0, 1, tmp.numel(), accumulate=True)
        return dq, dk, dvv, None


class _OffsetView:
    """a tensor seen from an element offset (only .data_ptr()/device/dtype are used by kern)."""

    def __init__(self, t: Tensor, off: int):
        self._t, self._off = t, off
        self.device, self.dtype, self.is_cuda = t.device, t.dtype, t.is_cuda

    def data_ptr(self):
        return self._t.data_ptr() + kern.esz(self._t) * self._off


def sr_attention(q, kv, heads):
    return SRAttentionFn.apply(q, kv, heads)


def nonlocal_attention(theta, phi, g):
    return NonlocalAttnFn.apply(theta, phi, g)


class NonlocalAttnJointFn(Function):
    """NonlocalAttnFn on theta | phi | g as the three channel thirds of ONE tensor [B, 3C, N] (the output of the single 1x1
    conv that ops.merged_param makes of conv_theta / conv_phi / conv_g, nlb.py:117-119): the kernels read the thirds in
    place (batch stride 3C N, element offsets 0 / C N / 2 C N) and write the three gradients into one [B, 3C, N] tensor, so
    that conv's backward is one data-gradient and one weight-gradient GEMM."""

    @staticmethod
    def forward(ctx, tpg):
        tpg = _c(tpg)
        B, C3 = tpg.shape[:2]
        Cn = C3 // 3
        N = tpg.numel() // (B * C3)
        ctx.tok64 = _bf(tpg) and Cn in (64, 128) and N >= 256
        ctx.dims = (B, Cn, N, tuple(tpg.shape))
        oshape = (B, Cn) + tuple(tpg.shape[2:])
        if ctx.tok64:
            # token-major copies of theta | phi | g as ONE tensor [B, 3, N, C] (one transpose launch over 3 B planes); the
            # attention kernels read the thirds in place (batch_mul = 3)
            T = torch.empty((B, 3, N, Cn), device=tpg.device, dtype=tpg.dtype)
            kern.transpose(tpg, Cn * N, T, Cn * N, 3 * B, Cn, N)
            U = torch.empty((B, 1, N, Cn), device=tpg.device, dtype=tpg.dtype)
            lse = torch.empty((B, 1, N), device=tpg.device, dtype=torch.float32)
            a = kern.DiffAttnT()
            e = T.element_size()
            a.q, a.k, a.v = T.data_ptr(), T.data_ptr() + e * N * Cn, T.data_ptr() + 2 * e * N * Cn
            a.U, a.lse = U.data_ptr(), lse.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale, a.batch_mul = B, 1, N, Cn, Cn ** -0.5, 3
            kern.attn64(a, backward=False)
            o = torch.empty(oshape, device=tpg.device, dtype=tpg.dtype)
            kern.transpose(U, N * Cn, o, Cn * N, B, N, Cn)
            ctx.save_for_backward(T, U, lse)
            return o
        st, so = (3 * Cn * N, 0, 1, N), (Cn * N, 0, 1, N)
        d = _AttnDesc(B, 1, N, N, Cn, Cn, Cn ** -0.5, 1, st, st, st, so, qoff=0, koff=Cn * N, voff=2 * Cn * N)
        o = torch.empty(oshape, device=tpg.device, dtype=tpg.dtype)
        kind, saved = _attn_forward(d, tpg, tpg, tpg, o)
        ctx.save_for_backward(tpg, o, saved)
        ctx.d, ctx.kind = d, kind
        return o

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        B, Cn, N, shape = ctx.dims
        dj = torch.empty(shape, device=g.device, dtype=g.dtype)
        if ctx.tok64:
            T, U, lse = ctx.saved_tensors
            gt = torch.empty_like(U)
            kern.transpose(g, Cn * N, gt, N * Cn, B, Cn, N)
            dT = torch.empty_like(T)  # dq | dk | dv, token-major, [B, 3, N, C]
            ws = torch.empty(kern.attn64_ws_bytes(B, 1, N), device=g.device, dtype=torch.uint8)
            a = kern.DiffAttnT()
            e = T.element_size()
            a.q, a.k, a.v = T.data_ptr(), T.data_ptr() + e * N * Cn, T.data_ptr() + 2 * e * N * Cn
            a.U, a.lse, a.dU = U.data_ptr(), lse.data_ptr(), gt.data_ptr()
            a.dq, a.dk, a.dv = dT.data_ptr(), dT.data_ptr() + e * N * Cn, dT.data_ptr() + 2 * e * N * Cn
            a.ws = ws.data_ptr()
            a.B, a.H, a.N, a.hd, a.scale, a.batch_mul = B, 1, N, Cn, Cn ** -0.5, 3
            kern.attn64(a, backward=True)
            kern.transpose(dT, N * Cn, dj, Cn * N, 3 * B, N, Cn)  # -> [B, 3 C, N] in one launch
            return dj
        tpg, o, saved = ctx.saved_tensors
        _attn_backward(ctx.d, ctx.kind, saved, tpg, tpg, tpg, o, g, dj, dj, dj)
        return dj


def nonlocal_attention_joint(tpg):
    """tpg [B, 3C, ...]: theta | phi | g stacked along the channel axis"""
    return NonlocalAttnJointFn.apply(tpg)


def diff_attention_heads(q, k, v, H):
    return DiffAttnHeadsFn.apply(q, k, v, H)


class DiffAttnCombineFn(Function):
    """multihead_diffattn.py:112-123: lambda, U[2h]-lambda U[2h+1], RMSNorm(2hd, eps 1e-5, no affine), *(1-lambda_init)."""

    @staticmethod
    def forward(ctx, U, lq1, lk1, lq2, lk2, lambda_init):
        U = _c(U)
        B, H2, N, dv = U.shape
        H = H2 // 2
        lam = _empty((3,), U)
        kern.diffattn_lambda_fwd(lq1, lk1, lq2, lk2, lambda_init, lam, lq1.numel())
        out = _act((B, N, H * dv), U)
        kern.diffattn_combine_fwd(U, lam, out, B, H, N, dv, 1e-5, 1.0 - lambda_init)
        ctx.save_for_backward(U, lam, lq1, lk1, lq2, lk2)
        ctx.refs = (lq1, lk1, lq2, lk2)
        ctx.lambda_init = lambda_init
        return out

    @staticmethod
    def backward(ctx, g):
        U, lam, lq1, lk1, lq2, lk2 = ctx.saved_tensors
        g = _c(g)
        B, H2, N, dv = U.shape
        H = H2 // 2
        dU = torch.empty_like(U)
        dlam = _zeros((1,), U)
        kern.diffattn_combine_bwd(U, lam, g, dU, dlam, B, H, N, dv, 1e-5, 1.0 - ctx.lambda_init)
        gs = [grad_buf(p) for p in ctx.refs]
        if gs[0] is not None:
            kern.diffattn_lambda_bwd(lq1, lk1, lq2, lk2, lam, dlam, gs[0], gs[1], gs[2], gs[3], lq1.numel())
        return dU, None, None, None, None, None


def diff_attention_combine(U, lq1, lk1, lq2, lk2, lambda_init):
    return DiffAttnCombineFn.apply(U, lq1, lk1, lq2, lk2, lambda_init)


__all__ = [n for n in dir() if not n.startswith("__")]
