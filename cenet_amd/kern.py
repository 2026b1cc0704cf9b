"""Thin, autograd-free Python bindings over the C ABI in include/cenet_hip.h.

Every function here launches hand-written HIP kernels on the current torch stream with raw data pointers.
Tensors must live on the GPU (the only exception is the tests' host-side SIMT checker, which swaps the library handle — see
cenet_amd/_lib.py).  Activation tensors are fp32 (parity mode) or bf16 (throughput mode); a call whose tensor arguments
include a bf16 tensor goes to the `_bf16` twin of the entry point (include/cenet_hip.h), in which every ACTIVATION pointer
is bf16 while parameters, statistics and parameter gradients stay fp32.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import EpiT, MatT

ACT = {"none": 0, "relu": 1, "lrelu": 2, "gelu": 3, "silu": 4, "sigmoid": 5}


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if isinstance(t, Ptr):
            t = t.t
        if t.dtype not in (torch.float32, torch.bfloat16, torch.int32, torch.int64, torch.uint8):
            raise TypeError(f"cenet_amd kernels take fp32 / bf16 tensors (int32 / int64 / uint8 for indices and masks); "
                            f"got {t.dtype}")
        if not t.is_cuda and not _lib.is_hostsim():
            raise RuntimeError("cenet_amd kernels run on the MI355X only: tensor is not on a CUDA/HIP device "
                               "(there is no CPU fallback)")


def P(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the current torch stream as a raw hipStream_t (this runs once per kernel launch: keep it cheap)"""
    if _lib.is_hostsim():
        return C.c_void_p(0)
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------------------
# GEMM / implicit GEMM
# ------------------------------------------------------------------------------------------------
BF16 = torch.bfloat16


def is_bf16(t) -> bool:
    return t is not None and t.dtype == BF16


def esz(t) -> int:
    """bytes per element of a tensor-like (tensors, ops._OffsetView)"""
    return 2 if t.dtype == BF16 else 4


def mat_plain(t: torch.Tensor, sr: int, sc: int, sb: int = 0, skb: int = 0, kfast: int = 0, offset: int = 0,
              sb2: int = 0, kinner: int = 0, sk_outer: int = 0) -> MatT:
    m = MatT()
    m.bf16 = t.dtype == BF16  # python-side attribute: picks cenet_gemm_bf16
    m.ptr = t.data_ptr() + esz(t) * offset
    m.sb, m.sb2, m.skb, m.sr, m.sc = sb, sb2, skb, sr, sc
    m.kinner, m.sk_outer = kinner, sk_outer
    m.mode, m.kfast = 0, kfast
    return m


def mat_im2col(t: torch.Tensor, *, sb: int, skb: int, sci: int, sy: int, sx: int, KH: int, KW: int, Pw: int,
               Hs: int, Ws: int, stride: int, pad: int, dil: int, patch_is_row: int, transposed: int,
               kfast: int, offset: int = 0) -> MatT:
    m = MatT()
    m.bf16 = t.dtype == BF16
    m.ptr = t.data_ptr() + esz(t) * offset
    m.sb, m.skb, m.sr, m.sc = sb, skb, 0, 0
    m.mode, m.kfast = 1, kfast
    m.patch_is_row, m.transposed = patch_is_row, transposed
    m.KH, m.KW, m.Pw, m.Hs, m.Ws = KH, KW, Pw, Hs, Ws
    m.stride, m.pad, m.dil = stride, pad, dil
    m.sci, m.sy, m.sx = sci, sy, sx
    return m


def pick_splits(M: int, N: int, nbatch: int, iters: int, target_blocks: int = 1024) -> int:
    blocks = ((M + 63) // 64) * ((N + 63) // 64) * nbatch
    if blocks >= target_blocks or iters <= 4:
        return 1
    return max(1, min(iters // 2, (target_blocks + blocks - 1) // blocks, 256))


def gemm(A: MatT, B: MatT, Cout: torch.Tensor, M: int, N: int, K: int, *, scr: int, scc: int, scb: int = 0,
         nbatch: int = 1, nkb: int = 1, splits: int = 1, bias: Optional[torch.Tensor] = None, bias_on_row: bool = False,
         act: str = "none", slope: float = 0.0, bscale: Optional[torch.Tensor] = None,
         R: Optional[torch.Tensor] = None, srb: int = 0, srr: int = 0, src: int = 0, atomic: bool = False,
         alpha: float = 1.0, c_offset: int = 0, r_offset: int = 0, nb_inner: int = 1, scb2: int = 0, srb2: int = 0,
         col2im: Optional[dict] = None, bscale_rows: int = 0, asum: Optional[torch.Tensor] = None):
    """asum (atomic contractions, nbatch == 1): fp32 vector that receives asum[m] += sum_k A[m, k] — the bias gradient riding
    in the weight-gradient pass"""
    _chk(Cout, bias, bscale, R, asum)
    if A.bf16 != B.bf16:
        raise TypeError("cenet_gemm: A and B must have the same element type (fp32 or bf16)")
    want = torch.float32 if (atomic or not A.bf16) else BF16  # atomic epilogues always add into fp32
    if Cout.dtype != want or (R is not None and R.dtype != want):
        raise TypeError(f"cenet_gemm: C / R must be {want} here (operands {'bf16' if A.bf16 else 'fp32'}, atomic={atomic}); "
                        f"got {Cout.dtype}" + (f" / {R.dtype}" if R is not None else ""))
    e = EpiT()
    e.C = Cout.data_ptr() + esz(Cout) * c_offset
    e.scb, e.scb2, e.scr, e.scc = scb, scb2, scr, scc
    e.bias = bias.data_ptr() if bias is not None else None
    e.bias_on_row = int(bias_on_row)
    e.act, e.slope = ACT[act], slope
    e.bscale = bscale.data_ptr() if bscale is not None else None
    e.bscale_rows = bscale_rows
    e.asum = asum.data_ptr() if asum is not None else None
    e.R = (R.data_ptr() + esz(R) * r_offset) if R is not None else None
    e.srb, e.srb2, e.srr, e.src = srb, srb2, srr, src
    e.atomic, e.alpha = int(atomic), alpha
    if col2im is not None:  # scatter epilogue (data-gradient of strided convolutions)
        e.cmode = 1
        e.cKH, e.cKW, e.cPw = col2im["KH"], col2im["KW"], col2im["Pw"]
        e.cHs, e.cWs, e.cstride, e.cpad = col2im["Hs"], col2im["Ws"], col2im["stride"], col2im["pad"]
        e.csci, e.csy, e.csx = col2im["sci"], col2im["sy"], col2im["sx"]
    fn = _lib.lib().cenet_gemm_bf16 if A.bf16 else _lib.lib().cenet_gemm_f32
    rc = fn(C.byref(A), C.byref(B), C.byref(e), M, N, K, nbatch, nb_inner, nkb, splits, stream())
    _lib.check(rc, "cenet_gemm_bf16" if A.bf16 else "cenet_gemm_f32")


class WgradProbT(C.Structure):
    """cenet_wgrad_prob_t (include/cenet_hip.h)."""
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("asum", C.c_void_p),
                ("lda", C.c_long), ("ldb", C.c_long), ("skbA", C.c_long), ("skbB", C.c_long),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("nkb", C.c_int), ("akf", C.c_int), ("bkf", C.c_int)]


def _wgrad_array(probs):
    n = len(probs)
    arr = (WgradProbT * n)()
    for d, t in zip(arr, probs):
        d.A, d.B, d.C, d.asum, d.lda, d.ldb, d.skbA, d.skbB, d.M, d.N, d.K, d.nkb = t[:12]
        d.akf = d.bkf = t[12]
    return arr


def wgrad_group(probs, device, phases=(0,), between=None):
    """probs: list of tuples (A_ptr, B_ptr, C_ptr, asum_ptr, lda, ldb, skbA, skbB, M, N, K, nkb, kfast) — all the recorded
    weight gradients of a backward segment in one call (gemm_group.hip): C_i += A_i^T-style contraction, asum_i += row sums.
    phases=(1, 2) issues the K-slice launches and the fold launches as two calls (`between()` runs in between): bench.py"""
    n = len(probs)
    arr = _wgrad_array(probs)
    lib = _lib.lib()
    f = lib.cenet_wgrad_group_ws_floats
    f.restype = C.c_long
    need = int(f(arr, n))
    ws = torch.empty(max(need, 1), device=device, dtype=torch.float32)
    for i, ph in enumerate(phases):
        if i and between is not None:
            between()
        if ph == 0:
            _lib.check(lib.cenet_wgrad_group_bf16(arr, n, P(ws), L(need), stream()), "cenet_wgrad_group_bf16")
        else:
            _lib.check(lib.cenet_wgrad_group_phase_bf16(arr, n, P(ws), L(need), int(ph), stream()), "cenet_wgrad_group_phase_bf16")
    return need


def wgrad_group_plan(probs):
    """[(launch index, bm, bn, ns)] per problem: the partition kern.wgrad_group makes (the library's own, gemm_group.hip)"""
    n = len(probs)
    arr = _wgrad_array(probs)
    out = [(C.c_int * n)() for _ in range(4)]
    _lib.check(_lib.lib().cenet_wgrad_group_plan(arr, n, *out), "cenet_wgrad_group_plan")
    return [tuple(int(o[i]) for o in out) for i in range(n)]


def last_gemm_kernel() -> str:
    """name of the kernel instance the last gemm() on this thread launched (as rocprofv3 prints it); measurement aid"""
    f = _lib.lib().cenet_gemm_last_kernel
    f.restype = C.c_char_p
    return f().decode()


# ------------------------------------------------------------------------------------------------
# generic call helper
# ------------------------------------------------------------------------------------------------
class Ptr:
    """a tensor seen from an element offset: Ptr(t, off) is passed to _call like a tensor"""

    def __init__(self, t, off: int = 0):
        self.t, self.off = t, off
        self.dtype = t.dtype

    def data_ptr(self):
        return self.t.data_ptr() + esz(self.t) * self.off


def _call(name, *args):
    """name ends in `_f32`; a bf16 tensor among the arguments selects the `_bf16` twin (all activation arguments of one call
    share their element type)"""
    cargs = []
    bf = False
    for a in args:
        if isinstance(a, (torch.Tensor, Ptr)):
            bf = bf or a.dtype == BF16
            cargs.append(C.c_void_p(a.data_ptr()))
        elif a is None:
            cargs.append(C.c_void_p(0))
        elif isinstance(a, float):
            cargs.append(C.c_float(a))
        elif isinstance(a, bool):
            cargs.append(C.c_int(int(a)))
        elif isinstance(a, int):
            cargs.append(C.c_long(a) if abs(a) > 0x7FFFFFFF else C.c_int(a))
        else:
            cargs.append(a)
    if bf and not name.endswith("_bf16"):  # (entries that exist for bf16 tensors only are named so by the caller)
        assert name.endswith("_f32"), name
        name = name[:-4] + "_bf16"
    rc = getattr(_lib.lib(), name)(*cargs, stream())
    _lib.check(rc, name)


def L(v: int):
    """force a C long argument"""
    return C.c_long(int(v))


class AttnT(C.Structure):
    """cenet_attn_t (include/cenet_hip.h)."""
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("delta", C.c_void_p)] + \
               [(n, C.c_long) for n in ("qsb", "qsh", "qsi", "qsd", "ksb", "ksh", "ksi", "ksd", "vsb", "vsh", "vsi", "vsd",
                                        "osb", "osh", "osi", "osd")] + \
               [(n, C.c_int) for n in ("B", "H", "Nq", "Nk", "D", "Dv", "v_head_div")] + [("scale", C.c_float)] + \
               [("dkv_zeroed", C.c_int), ("dkv_f32", C.c_int), ("finite_scores", C.c_int)]


def flash_supported(D: int, Dv: int) -> bool:
    return bool(_lib.lib().cenet_flash_attn_supported(int(D), int(Dv)))


def flashb_supported(D: int, Dv: int) -> bool:
    """head dims the bf16 (throughput-mode) tiled attention kernels cover: D <= 64, Dv <= 128"""
    return D <= 64 and Dv <= 128


def flash_fwd(a: AttnT, bf16: bool = False):
    name = "cenet_flash_attn_fwd_bf16" if bf16 else "cenet_flash_attn_fwd_f32"
    _lib.check(getattr(_lib.lib(), name)(C.byref(a), stream()), name)


def flash_bwd(a: AttnT, bf16: bool = False):
    name = "cenet_flash_attn_bwd_bf16" if bf16 else "cenet_flash_attn_bwd_f32"
    _lib.check(getattr(_lib.lib(), name)(C.byref(a), stream()), name)


class DiffAttnT(C.Structure):
    """cenet_diffattn_t (include/cenet_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("q", "k", "v", "U", "lse", "dU", "dq", "dk", "dv", "ws")] + \
               [(n, C.c_int) for n in ("B", "H", "N", "hd")] + [("scale", C.c_float), ("batch_mul", C.c_int)]


def diffattn_heads_supported(hd: int, N: int) -> bool:
    return bool(_lib.lib().cenet_diffattn_heads_supported(int(hd), int(N)))


def diffattn_heads_ws_bytes(B: int, H: int, N: int) -> int:
    fn = _lib.lib().cenet_diffattn_heads_ws_bytes
    fn.restype = C.c_long
    return int(fn(int(B), int(H), int(N)))


def diffattn_heads(a: DiffAttnT, backward: bool):
    name = "cenet_diffattn_heads_bwd_bf16" if backward else "cenet_diffattn_heads_fwd_bf16"
    _lib.check(getattr(_lib.lib(), name)(C.byref(a), stream()), name)


def attn64_ws_bytes(B: int, H: int, N: int) -> int:
    fn = _lib.lib().cenet_attn64_ws_bytes
    fn.restype = C.c_long
    return int(fn(int(B), int(H), int(N)))


def attn64(a: DiffAttnT, backward: bool):
    """plain self-attention, head dimension 64, bf16 token-major tensors (attn_diff.hip, single-softmax form)"""
    name = "cenet_attn64_bwd_bf16" if backward else "cenet_attn64_fwd_bf16"
    _lib.check(getattr(_lib.lib(), name)(C.byref(a), stream()), name)


def sra_attn_bwd_supported(hd: int, Nk: int) -> bool:
    return bool(_lib.lib().cenet_sra_attn_bwd_supported(int(hd), int(Nk)))


def sra_attn_bwd_blocks_supported(hd: int, Nk: int) -> bool:
    return bool(_lib.lib().cenet_sra_attn_bwd_blocks_supported(int(hd), int(Nk)))


def sra_attn_fwd(q, kv, o, lse, B, H, Nq, Nk, scale):
    """spatial-reduction attention forward with resident keys (bf16, head dim 64, <= 64 keys): o and the natural-log lse"""
    _chk(q, kv, o, lse)
    assert q.dtype == BF16 and kv.dtype == BF16 and o.dtype == BF16 and lse.dtype == torch.float32
    rc = _lib.lib().cenet_sra_attn_fwd_bf16(P(q), P(kv), P(o), P(lse), B, H, Nq, Nk, C.c_float(scale), stream())
    _lib.check(rc, "cenet_sra_attn_fwd_bf16")


def sra_attn_bwd(q, kv, o, dout, lse, dq, dkv, B, H, Nq, Nk, scale):
    """fused spatial-reduction attention backward (bf16, head dim 64, <= 64 keys); dkv: fp32, zero-filled"""
    _chk(q, kv, o, dout, lse, dq, dkv)
    assert q.dtype == BF16 and kv.dtype == BF16 and dq.dtype == BF16 and dkv.dtype == torch.float32
    rc = _lib.lib().cenet_sra_attn_bwd_bf16(P(q), P(kv), P(o), P(dout), P(lse), P(dq), P(dkv), B, H, Nq, Nk, C.c_float(scale),
                                            stream())
    _lib.check(rc, "cenet_sra_attn_bwd_bf16")


def sra_attn_bwd_direct_supported(B, H, Nq, Nk) -> bool:
    return bool(_lib.lib().cenet_sra_attn_bwd_direct_supported(int(B), int(H), int(Nq), int(Nk)))


def sra_attn_bwd_direct(q, kv, o, dout, lse, dq, dkv, B, H, Nq, Nk, scale):
    """as sra_attn_bwd where one workgroup owns a (batch, head): dkv bf16, written (not accumulated)"""
    _chk(q, kv, o, dout, lse, dq, dkv)
    assert q.dtype == BF16 and kv.dtype == BF16 and dq.dtype == BF16 and dkv.dtype == BF16
    rc = _lib.lib().cenet_sra_attn_bwd_direct_bf16(P(q), P(kv), P(o), P(dout), P(lse), P(dq), P(dkv), B, H, Nq, Nk,
                                                   C.c_float(scale), stream())
    _lib.check(rc, "cenet_sra_attn_bwd_direct_bf16")


def softmax_rows_fwd(x, y, rows, n):
    _chk(x, y)
    _call("cenet_softmax_rows_fwd_f32", x, y, L(rows), n)


def softmax_rows_bwd(y, dy, dx, rows, n):
    _chk(y, dy, dx)
    _call("cenet_softmax_rows_bwd_f32", y, dy, dx, L(rows), n)


# ---- norms -----------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, Cn, eps):
    _chk(x, gamma, beta, y, mean, rstd)
    _call("cenet_layernorm_fwd_f32", x, gamma, beta, y, mean, rstd, rows, Cn, float(eps))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, Cn, dx_add=None):
    _chk(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dx_add)
    _call("cenet_layernorm_bwd_add_acc_f32", dy, x, gamma, mean, rstd, dx_add, dx, dgamma, dbeta, rows, Cn)


def layernorm_bwd_part_supported(dy, x, Cn) -> bool:
    return bool(is_bf16(dy) and is_bf16(x) and Cn % 8 == 0 and Cn <= 512 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0)


def layernorm_bwd_part(dy, x, gamma, mean, rstd, dx, rows, Cn, dx_add=None, bscale=None, dxs=None):
    """LayerNorm backward whose affine gradients go to a partial buffer [workgroups, 2 C] (returned) instead of float atomics;
    ln_fold_group adds the column sums of many such buffers into their gradients with one launch.
    bscale [B] + dxs (like dx): also write dxs = bscale[sample] * dx (what scale_batch(dx, bscale) would)."""
    _chk(dy, x, gamma, mean, rstd, dx, dx_add, bscale, dxs)
    nrows = int(_lib.lib().cenet_layernorm_bwd_part_rows(rows, Cn))
    part = torch.empty((nrows, 2 * Cn), device=dy.device, dtype=torch.float32)
    if bscale is None:
        _call("cenet_layernorm_bwd_add_part_bf16", dy, x, gamma, mean, rstd, dx_add, dx, part, rows, Cn)
    else:
        _call("cenet_layernorm_bwd_add_part_scaled_bf16", dy, x, gamma, mean, rstd, dx_add, dx, part, bscale,
              rows // bscale.numel(), dxs, rows, Cn)
    return part


def ln_fold_group(items):
    """items: [(part [nrows, 2 C], dgamma [C], dbeta [C])]: dgamma / dbeta += column sums, one launch per 48 items"""
    n = len(items)
    if n == 0:
        return
    parts = (C.c_void_p * n)(*[t[0].data_ptr() for t in items])
    dgs = (C.c_void_p * n)(*[t[1].data_ptr() for t in items])
    dbs = (C.c_void_p * n)(*[t[2].data_ptr() for t in items])
    nr = (C.c_int * n)(*[t[0].shape[0] for t in items])
    cs = (C.c_int * n)(*[t[0].shape[1] // 2 for t in items])
    rc = _lib.lib().cenet_ln_fold_group(parts, dgs, dbs, nr, cs, n, stream())
    _lib.check(rc, "cenet_ln_fold_group")


def bn_stats(x, sb, B, Cn, HW, ws, mean, var, rmean, rvar, momentum, nbt):
    _chk(x, ws, mean, var, rmean, rvar)
    _call("cenet_bn_stats_f32", x, L(sb), B, Cn, HW, ws, mean, var, rmean, rvar, float(momentum), nbt)


def bn1d_supported(B: int) -> bool:
    return bool(_lib.lib().cenet_bn1d_supported(int(B)))


def bn1d_train_fwd(z, zn, mean, var, rmean, rvar, momentum, nbt, eps, gamma, beta, B, Cn):
    """train-mode BatchNorm1d of a [B, C] fp32 matrix (2 <= B <= 64) in one launch: zn, mean / var, running statistics, counter"""
    _chk(z, zn, mean, var, rmean, rvar, gamma, beta)
    assert z.dtype == torch.float32 and zn.dtype == torch.float32
    _lib.check(_lib.lib().cenet_bn1d_train_fwd_f32(P(z), P(zn), P(mean), P(var), P(rmean), P(rvar), C.c_float(momentum), P(nbt),
                                                   C.c_float(eps), P(gamma), P(beta), int(B), int(Cn), stream()),
               "cenet_bn1d_train_fwd_f32")


def bn1d_bwd(dy, z, dz, mean, var, eps, gamma, dgamma, dbeta, B, Cn):
    _chk(dy, z, dz, mean, var, gamma, dgamma, dbeta)
    assert dy.dtype == torch.float32 and z.dtype == torch.float32
    _lib.check(_lib.lib().cenet_bn1d_bwd_acc_f32(P(dy), P(z), P(dz), P(mean), P(var), C.c_float(eps), P(gamma), P(dgamma), P(dbeta),
                                                 int(B), int(Cn), stream()), "cenet_bn1d_bwd_acc_f32")


def bn_train_fwd(x, sxb, y, syb, ws, mean, var, rmean, rvar, momentum, nbt, eps, gamma, beta, act, slope, B, Cn, HW):
    """train-mode BatchNorm forward: statistics + normalisation (+ activation), mean / var / running statistics written"""
    _chk(x, y, ws, mean, var, rmean, rvar, gamma, beta)
    _call("cenet_bn_train_fwd_f32", x, L(sxb), y, L(syb), ws, mean, var, rmean, rvar, float(momentum), nbt, float(eps), gamma,
          beta, ACT[act], float(slope), B, Cn, HW, int(nbt.numel()) if nbt is not None else 1)


def bn_apply(x, sxb, y, syb, mean, var, eps, gamma, beta, act, slope, B, Cn, HW):
    _chk(x, y, mean, var, gamma, beta)
    _call("cenet_bn_apply_f32", x, L(sxb), y, L(syb), mean, var, float(eps), gamma, beta, ACT[act], float(slope), B, Cn, HW)


def bn_bwd(dy, sgb, x, sxb, dx, sdb, mean, var, eps, gamma, beta, act, slope, B, Cn, HW, ws, dgamma, dbeta, dx_add=None):
    """dx_add (contiguous [B, C, HW], optional): added to dx by the kernel that writes it (the gradient of a residual connection
    around the BatchNorm)"""
    _chk(dy, x, dx, mean, var, gamma, beta, ws, dgamma, dbeta, dx_add)
    if dx_add is None:
        _call("cenet_bn_bwd_acc_f32", dy, L(sgb), x, L(sxb), dx, L(sdb), mean, var, float(eps), gamma, beta, ACT[act],
              float(slope), B, Cn, HW, ws, dgamma, dbeta)
    else:
        _call("cenet_bn_bwd_add_acc_f32", dy, L(sgb), x, L(sxb), dx, L(sdb), dx_add, L(Cn * HW), mean, var, float(eps), gamma, beta,
              ACT[act], float(slope), B, Cn, HW, ws, dgamma, dbeta)


def res_tail_supported(x2, x3) -> bool:
    return bool(is_bf16(x2) and is_bf16(x3) and x2.dim() == 4 and x2.shape == x3.shape and x2.data_ptr() % 16 == 0
                and x3.data_ptr() % 16 == 0 and _lib.lib().cenet_res_tail_supported(int(x2.shape[2]), int(x2.shape[3])))


def res_tail_fwd(x2, x3, mean2, var2, g2, b2, eps2, mean3, var3, g3, b3, eps3, w, slope, out, B, Cn, H, W):
    """out = w[c] * MaxPool2x2(LeakyReLU(BN2(x2) + BN3(x3))) from the batch statistics, one launch (csrc/res_tail.hip)"""
    _chk(x2, x3, mean2, var2, g2, b2, mean3, var3, g3, b3, w, out)
    _call("cenet_res_tail_fwd_bf16", x2, x3, mean2, var2, g2, b2, float(eps2), mean3, var3, g3, b3, float(eps3), w, float(slope), out,
          B, Cn, H, W)


def res_tail_bwd(g, x2, x3, mean2, var2, g2, b2, eps2, mean3, var3, g3, b3, eps3, w, slope, dx2, dx3, dg2, db2, dg3, db3, dw, B, Cn,
                 H, W):
    _chk(g, x2, x3, mean2, var2, g2, b2, mean3, var3, g3, b3, w, dx2, dx3, dg2, db2, dg3, db3, dw)
    f = _lib.lib().cenet_res_tail_bwd_ws_floats
    f.restype = C.c_long
    ws = torch.empty(int(f(Cn)), device=g.device, dtype=torch.float32)
    _call("cenet_res_tail_bwd_bf16", g, x2, x3, mean2, var2, g2, b2, float(eps2), mean3, var3, g3, b3, float(eps3), w, float(slope),
          dx2, dx3, dg2, db2, dg3, db3, dw, ws, B, Cn, H, W)


def res_tail_img_fwd(x2, img, mean2, var2, g2, b2, eps2, imean, ivar, w3, g3, b3, eps3, rm3, rv3, nbt3, mom3, w, slope, out, B, Cn, H, W):
    """res_tail_fwd with the shortcut x3 = w3[c] * img not materialised (one-channel input); updates BatchNorm3's running statistics"""
    _chk(x2, img, mean2, var2, g2, b2, imean, ivar, w3, g3, b3, rm3, rv3, w, out)
    _call("cenet_res_tail_img_fwd_bf16", x2, img, mean2, var2, g2, b2, float(eps2), imean, ivar, w3, g3, b3, float(eps3), rm3, rv3, nbt3,
          float(mom3), w, float(slope), out, B, Cn, H, W)


def res_tail_img_bwd(g, x2, img, mean2, var2, g2, b2, eps2, imean, ivar, w3, g3, b3, eps3, w, slope, dx2, dg2, db2, dg3, db3, dw3, dw,
                     B, Cn, H, W):
    _chk(g, x2, img, mean2, var2, g2, b2, imean, ivar, w3, g3, b3, w, dx2, dg2, db2, dg3, db3, dw3, dw)
    f = _lib.lib().cenet_res_tail_bwd_ws_floats
    f.restype = C.c_long
    ws = torch.empty(int(f(Cn)), device=g.device, dtype=torch.float32)
    _call("cenet_res_tail_img_bwd_bf16", g, x2, img, mean2, var2, g2, b2, float(eps2), imean, ivar, w3, g3, b3, float(eps3), w,
          float(slope), dx2, dg2, db2, dg3, db3, dw3, dw, ws, B, Cn, H, W)


# ---- depthwise conv ------------------------------------------------------------------------------------
# ---- channel-local fused chains (csrc/chanloc.hip) --------------------------------------------------------------------------
_NO_CHANLOC = __import__("os").environ.get("CENET_NO_CHANLOC") is not None  # measurement aid: the unfused launch chains


def eucb_supported(x) -> bool:
    B, _, H, W = x.shape
    return (not _NO_CHANLOC) and bool(_lib.lib().cenet_eucb_supported(int(B), int(H), int(W), esz(x)))


def eucb_fwd(x, w, gamma, beta, eps, slope, y, mean, var, rmean, rvar, momentum, nbt, B, Cn, H, W):
    _chk(x, w, gamma, beta, y, mean, var, rmean, rvar)
    _call("cenet_eucb_fwd_f32", x, L(Cn * H * W), w, gamma, beta, float(eps), float(slope), y, L(4 * Cn * H * W), mean, var, rmean,
          rvar, float(momentum), nbt, B, Cn, H, W)


def eucb_bwd(g, x, w, gamma, beta, eps, slope, mean, var, dx, dw, dgamma, dbeta, B, Cn, H, W):
    _chk(g, x, w, gamma, beta, mean, var, dx, dw, dgamma, dbeta)
    _call("cenet_eucb_bwd_acc_f32", g, L(4 * Cn * H * W), x, L(Cn * H * W), w, gamma, beta, float(eps), float(slope), mean, var, dx,
          L(Cn * H * W), dw, dgamma, dbeta, B, Cn, H, W)


def chanloc_supported(B: int, HW: int) -> bool:
    return (not _NO_CHANLOC) and bool(_lib.lib().cenet_chanloc_supported(int(B), int(HW)))


def cfam_mid_fwd(p_raw, m, x0, x1, y2, gp, bp, epsp, meanp, varp, rmp, rvp, momp, nbtp, w, ls, g2, b2, eps2, mean2, var2, rm2, rv2,
                 mom2, nbt2, B, Cn, HW):
    _chk(p_raw, m, x0, x1, y2, gp, bp, meanp, varp, w, ls, g2, b2, mean2, var2)
    _call("cenet_cfam_mid_fwd_f32", p_raw, m, x0, x1, y2, gp, bp, float(epsp), meanp, varp, rmp, rvp, float(momp), nbtp, w, ls, g2,
          b2, float(eps2), mean2, var2, rm2, rv2, float(mom2), nbt2, B, Cn, HW)


def cfam_mid_bwd(g_y2, g_x1, p_raw, m, x1, d_p, d_m, d_x0, gp, bp, epsp, meanp, varp, w, ls, g2, eps2, mean2, var2, dgp, dbp, dw, dls,
                 dg2, db2, B, Cn, HW):
    _chk(g_y2, g_x1, p_raw, m, x1, d_p, d_m, d_x0)
    _call("cenet_cfam_mid_bwd_acc_f32", g_y2, g_x1, p_raw, m, x1, d_p, d_m, d_x0, gp, bp, float(epsp), meanp, varp, w, ls, g2,
          float(eps2), mean2, var2, dgp, dbp, dw, dls, dg2, db2, B, Cn, HW)


def dwbn_fwd(x, ws, dils, g, p, v, rest, gamma, beta, eps, mean, var, rmean, rvar, momentum, nbt, B, H, W):
    """MultiOrderDWConv's dilated depthwise branches + BatchNorm + ReLU and the pooled slice's copy in one launch (chanloc.hip)"""
    _chk(x, v, rest, gamma, beta, mean, var, rmean, rvar, *ws)
    nb, Ct = len(ws), len(ws) * g + p
    _call("cenet_dwbn_fwd_f32", x, L(Ct * H * W), _ptr_arr([w.data_ptr() for w in ws]), (C.c_int * nb)(*[int(d) for d in dils]), nb,
          g, p, v, L(nb * g * H * W), rest, L(p * H * W), gamma, beta, float(eps), mean, var, rmean, rvar, float(momentum), nbt, B, H, W)


def dwbn_bwd(g_v, g_rest, x, ws, dils, g, p, gamma, beta, eps, mean, var, dx, dws, dgamma, dbeta, B, H, W):
    _chk(g_v, g_rest, x, dx, dgamma, dbeta, *ws, *dws)
    nb, Ct = len(ws), len(ws) * g + p
    _call("cenet_dwbn_bwd_acc_f32", g_v, L(nb * g * H * W), g_rest, L(p * H * W), None, L(0), x, L(Ct * H * W),
          _ptr_arr([w.data_ptr() for w in ws]), (C.c_int * nb)(*[int(d) for d in dils]), nb, g, p, gamma, beta, float(eps), mean, var,
          dx, L(Ct * H * W), _ptr_arr([w.data_ptr() for w in dws]), dgamma, dbeta, B, H, W)


def cfam_front_fwd(x0, y1, xs, g1, b1, eps1, mean1, var1, rm1, rv1, mom1, nbt1, fc1, fc2, gd, bd, epsd, meand, vard, rmd, rvd, momd,
                   nbtd, u, amax, z, zn, B, Cn, HW):
    _chk(x0, y1, xs, g1, b1, mean1, var1, fc1, fc2, u, amax, z, zn)
    _call("cenet_cfam_front_fwd_f32", x0, y1, xs, g1, b1, float(eps1), mean1, var1, rm1, rv1, float(mom1), nbt1, fc1, fc2, gd, bd,
          float(epsd), meand, vard, rmd, rvd, float(momd), nbtd, u, amax, z, zn, B, Cn, HW)


def cfam_front_bwd(g_xs, g_y1, g_tap, x0, dx0, g1, b1, eps1, mean1, var1, fc1, fc2, gd, epsd, meand, vard, u, amax, z, zn, dg1, db1,
                   dfc1, dfc2, dgd, dbd, B, Cn, HW):
    _chk(g_xs, g_y1, g_tap, x0, dx0)
    _call("cenet_cfam_front_bwd_acc_f32", g_xs, g_y1, g_tap, x0, dx0, g1, b1, float(eps1), mean1, var1, fc1, fc2, gd, float(epsd),
          meand, vard, u, amax, z, zn, dg1, db1, dfc1, dfc2, dgd, dbd, B, Cn, HW)


def dwact_supported(x) -> bool:
    B, _, H, W = x.shape
    return B * H * W <= 8192 and chanloc_supported(B, H * W)


def dwact_fwd(x, w, bias, y, act, slope, dil, B, Cn, H, W):
    _chk(x, w, bias, y)
    _call("cenet_dwact_fwd_f32", x, w, bias, y, ACT[act], float(slope), dil, B, Cn, H, W)


def dwact_bwd(g, x, w, bias, dx, dw, db, act, slope, dil, B, Cn, H, W):
    _chk(g, x, w, bias, dx, dw, db)
    _call("cenet_dwact_bwd_acc_f32", g, x, w, bias, dx, dw, db, ACT[act], float(slope), dil, B, Cn, H, W)


def pool_branch_supported(B, P, H, W) -> bool:
    return (not _NO_CHANLOC) and 1 <= B <= 64 and 1 <= P <= 32 and 1 <= H <= 64 and 1 <= W <= 64 and H * W >= 16


def pool_branch_fwd(x, sxb, wc, gamma, beta, eps, slope, RH, RW, y, syb, pooled, t, mean, var, rmean, rvar, momentum, nbt, B, P, H, W):
    _chk(x, wc, gamma, beta, RH, RW, y, pooled, t, mean, var, rmean, rvar)
    _call("cenet_pool_branch_fwd_f32", x, L(sxb), wc, gamma, beta, float(eps), float(slope), RH, RW, y, L(syb), pooled, t, mean, var,
          rmean, rvar, float(momentum), nbt, B, P, H, W)


def pool_branch_bwd(g, sgb, wc, gamma, beta, eps, slope, RH, RW, pooled, t, mean, var, dt_ws, dx, sdb, dwc, dgamma, dbeta, B, P, H, W):
    _chk(g, wc, gamma, beta, RH, RW, pooled, t, mean, var, dt_ws, dx, dwc, dgamma, dbeta)
    _call("cenet_pool_branch_bwd_acc_f32", g, L(sgb), wc, gamma, beta, float(eps), float(slope), RH, RW, pooled, t, mean, var, dt_ws,
          dx, L(sdb), dwc, dgamma, dbeta, B, P, H, W)


def dw_nchw(x, sxb, w, bias, y, syb, a, sab, B, Cn, H, W, dil, flip, act="none", slope=0.0, x_off=0, y_off=0):
    """x_off / y_off (elements): read / write a channel slice of a wider tensor in place (batch strides sxb / syb)"""
    _chk(x, w, bias, y, a)
    _call("cenet_dwconv3x3_nchw_f32", Ptr(x, x_off), L(sxb), w, bias, Ptr(y, y_off) if y is not None else None, L(syb), a, L(sab),
          B, Cn, H, W, dil, int(flip), ACT[act], float(slope))



_NO_DW_MULTI = __import__("os").environ.get("CENET_DW_NO_MULTI") is not None  # measurement aid: the branches one launch each


def _ptr_arr(vals):
    return (C.c_void_p * len(vals))(*vals)


def dw_nchw_multi(branches, B, H, W, flip):
    """branches: [(x, x_off, sxb, w, y, y_off, syb, C, dil)] — up to 4 bias-free depthwise 3x3 convs (or their data gradients, flip)
    of bf16 NCHW channel slices in ONE launch.  Returns False when the library has no single-launch form for them (the caller
    then launches them one by one)."""
    n = len(branches)
    if n < 1 or n > 4 or _NO_DW_MULTI or any(not is_bf16(b[0]) or not is_bf16(b[4]) for b in branches):
        return False
    _chk(*[b[0] for b in branches], *[b[3] for b in branches], *[b[4] for b in branches])
    xs = _ptr_arr([b[0].data_ptr() + 2 * b[1] for b in branches])
    ws = _ptr_arr([b[3].data_ptr() for b in branches])
    ys = _ptr_arr([b[4].data_ptr() + 2 * b[5] for b in branches])
    sx = (C.c_long * n)(*[b[2] for b in branches])
    sy = (C.c_long * n)(*[b[6] for b in branches])
    cs = (C.c_int * n)(*[b[7] for b in branches])
    ds = (C.c_int * n)(*[b[8] for b in branches])
    rc = _lib.lib().cenet_dwconv3x3_nchw_multi_bf16(xs, sx, ws, ys, sy, cs, ds, n, B, H, W, int(flip), stream())
    if rc == 2:  # CENET_EUNSUPPORTED
        return False
    _lib.check(rc, "cenet_dwconv3x3_nchw_multi_bf16")
    return True


def dw_wgrad_nchw_multi(branches, B, H, W):
    """branches: [(x, x_off, sxb, dy, g_off, sgb, dw, C, dil)]: dw[i] += dy[i] (*) x[i] for up to 4 branches in ONE launch; False
    when unsupported"""
    n = len(branches)
    if n < 1 or n > 4 or _NO_DW_MULTI or any(not is_bf16(b[0]) or not is_bf16(b[3]) for b in branches):
        return False
    _chk(*[b[0] for b in branches], *[b[3] for b in branches], *[b[6] for b in branches])
    xs = _ptr_arr([b[0].data_ptr() + 2 * b[1] for b in branches])
    gs = _ptr_arr([b[3].data_ptr() + 2 * b[4] for b in branches])
    dws = _ptr_arr([b[6].data_ptr() for b in branches])
    sx = (C.c_long * n)(*[b[2] for b in branches])
    sg = (C.c_long * n)(*[b[5] for b in branches])
    cs = (C.c_int * n)(*[b[7] for b in branches])
    ds = (C.c_int * n)(*[b[8] for b in branches])
    rc = _lib.lib().cenet_dwconv3x3_wgrad_nchw_multi_bf16(xs, sx, gs, sg, dws, cs, ds, n, B, H, W, stream())
    if rc == 2:
        return False
    _lib.check(rc, "cenet_dwconv3x3_wgrad_nchw_multi_bf16")
    return True

def dw_tok(x, w, bias, y, a, B, Cn, H, W, flip, act="none", slope=0.0):
    _chk(x, w, bias, y, a)
    _call("cenet_dwconv3x3_tok_f32", x, w, bias, y, a, B, Cn, H, W, int(flip), ACT[act], float(slope))


def dw_tok_bwd_pre(x, g, w, bias, gu, dw, dbias, B, Cn, H, W, act, slope=0.0):
    """bf16 tokens: gu = g * act'(DW3x3(x) + bias), dw += gu (*) x, dbias += sum gu (pre-activation recomputed from x)."""
    _chk(x, g, w, bias, gu, dw, dbias)
    assert is_bf16(x) and is_bf16(g) and is_bf16(gu)
    _call("cenet_dwconv3x3_tok_bwd_pre_bf16", x, g, w, bias, gu, dw, dbias, B, Cn, H, W, ACT[act], float(slope))


def dw_tok_tiled(x):
    """the LDS-tiled bf16 token-layout depthwise kernels apply (16-byte channel groups)"""
    return is_bf16(x) and x.shape[-1] % 8 == 0 and x.data_ptr() % 16 == 0



def pvt_mlp_supported(x, Cn, HD, H, W) -> bool:
    """the fused PVT-MLP kernels (pvt_mlp.hip) have an instance for these bf16 tokens"""
    return bool(is_bf16(x) and x.data_ptr() % 16 == 0 and _lib.lib().cenet_pvt_mlp_supported(Cn, HD, H, W))


def pvt_mlp_fwd(x, ln_g, ln_b, eps, w1, b1, wd, bd, w2, b2, bscale, y, B, H, W, Cn, HD, saved=None):
    """y = x + s_b (fc2(GELU(DW3x3(fc1(LN(x))) + bd)) + b2), one launch (bf16 tokens; w1 / w2 are the bf16 shadows).
    saved = (xn, mean, rstd, h, a): also store the LayerNorm output and statistics, the fc1 output and s_b * the GELU output (what
    the backward kernels and the fc1 / fc2 weight gradients read)."""
    xn, mean, rstd, h, a = saved if saved is not None else (None,) * 5
    _chk(x, ln_g, ln_b, w1, b1, wd, bd, w2, b2, bscale, y, xn, mean, rstd, h, a)
    assert is_bf16(x) and is_bf16(y) and is_bf16(w1) and is_bf16(w2)
    _call("cenet_pvt_mlp_fwd_bf16", x, ln_g, ln_b, float(eps), w1, b1, wd, bd, w2, b2, bscale, y, xn, mean, rstd, h, a, B, H, W,
          Cn, HD)



def pvt_mlp_bwd(g, bscale, w1, w2, wd, bd, h, x, ln_g, mean, rstd, gu, dh, dx, dwd, dbd, dln_g, dln_b, db2, B, H, W, Cn, HD,
                up_scale=None, dxs=None):
    """backward of pvt_mlp_fwd from its saved tensors in two launches (+ a fold): gu = (s_b g . W2) * GELU'(DW(h) + bd) with the
    depthwise weight / bias gradients ADDED into dwd / dbd; dh = DW^T(gu) (the operand of the fc1 weight gradient), dx = g +
    LayerNormBackward(dh . W1) with the affine gradients ADDED into dln_g / dln_b and the fc2 bias gradient (column sums of s_b g)
    into db2 (None: not wanted).  up_scale [B] + dxs: also dxs = up_scale_b * dx (see ops._prescaled_put)."""
    _chk(g, bscale, w1, w2, wd, bd, h, x, ln_g, mean, rstd, gu, dh, dx, dwd, dbd, dln_g, dln_b, db2, up_scale, dxs)
    f = _lib.lib().cenet_pvt_mlp_bwd_ws_floats
    f.restype = C.c_long
    ws = torch.empty(int(f(B, H, W, Cn)), device=g.device, dtype=torch.float32)
    _call("cenet_pvt_mlp_bwd_bf16", g, bscale, w1, w2, wd, bd, h, x, ln_g, mean, rstd, gu, dh, dx, up_scale, dxs, dwd, dbd, dln_g,
          dln_b, db2, ws, B, H, W, Cn, HD)


def dw_wgrad_nchw(x, sxb, dy, sgb, dw, dbias, B, Cn, H, W, dil, x_off=0, g_off=0):
    _chk(x, dy, dw, dbias)
    _call("cenet_dwconv3x3_wgrad_nchw_acc_f32", Ptr(x, x_off), L(sxb), Ptr(dy, g_off), L(sgb), dw, dbias, B, Cn, H, W, dil)


def dw_wgrad_tok(x, dy, dw, dbias, B, Cn, H, W):
    _chk(x, dy, dw, dbias)
    _call("cenet_dwconv3x3_wgrad_tok_acc_f32", x, dy, dw, dbias, B, Cn, H, W)


# ---- resampling ------------------------------------------------------------------------------------------
def bilinear_fwd(x, sxb, y, syb, B, Cn, Hi, Wi, Ho, Wo, sh, sw, align):
    _chk(x, y)
    _call("cenet_bilinear_fwd_f32", x, L(sxb), y, L(syb), B, Cn, Hi, Wi, Ho, Wo, float(sh), float(sw), int(align))


def bilinear_bwd(dy, sgb, dx, sdb, B, Cn, Hi, Wi, Ho, Wo, sh, sw, align, dx_add=None):
    """dx_add (laid out like dx, optional): added to dx by the kernel that writes it"""
    _chk(dy, dx, dx_add)
    if dx_add is None:
        _call("cenet_bilinear_bwd_f32", dy, L(sgb), dx, L(sdb), B, Cn, Hi, Wi, Ho, Wo, float(sh), float(sw), int(align))
    else:
        _call("cenet_bilinear_bwd_add_f32", dy, L(sgb), dx, L(sdb), dx_add, B, Cn, Hi, Wi, Ho, Wo, float(sh), float(sw), int(align))


def nearest2x_fwd(x, sxb, y, syb, B, Cn, Hi, Wi):
    _chk(x, y)
    _call("cenet_nearest2x_fwd_f32", x, L(sxb), y, L(syb), B, Cn, Hi, Wi)


def nearest2x_bwd(dy, sgb, dx, sdb, B, Cn, Hi, Wi):
    _chk(dy, dx)
    _call("cenet_nearest2x_bwd_f32", dy, L(sgb), dx, L(sdb), B, Cn, Hi, Wi)


def avgpool_fwd(x, sxb, y, syb, B, Cn, Hi, Wi, Ho, Wo):
    _chk(x, y)
    _call("cenet_adaptive_avgpool_fwd_f32", x, L(sxb), y, L(syb), B, Cn, Hi, Wi, Ho, Wo)


def avgpool_bwd(dy, sgb, dx, sdb, B, Cn, Hi, Wi, Ho, Wo):
    _chk(dy, dx)
    _call("cenet_adaptive_avgpool_bwd_f32", dy, L(sgb), dx, L(sdb), B, Cn, Hi, Wi, Ho, Wo)


def maxpool2_fwd(x, y, syb, scale, B, Cn, Hi, Wi):
    _chk(x, y, scale)
    _call("cenet_maxpool2_fwd_f32", x, y, L(syb), scale, B, Cn, Hi, Wi)


def maxpool2_bwd(x, dy, sgb, dx, scale, dscale, B, Cn, Hi, Wi):
    _chk(x, dy, dx, scale, dscale)
    _call("cenet_maxpool2_bwd_acc_f32", x, dy, L(sgb), dx, scale, dscale, B, Cn, Hi, Wi)


# ---- CCU / SRM ---------------------------------------------------------------------------------------------
def ccu_stats_fwd(x, fc1, fc2, u, amax, z, B, Cn, HW):
    _chk(x, fc1, fc2, u, amax, z)
    _call("cenet_ccu_stats_fwd_f32", x, fc1, fc2, u, amax, z, B, Cn, HW)


def gate_chan_fwd(x, g, y, BC, HW):
    _chk(x, g, y)
    _call("cenet_gate_chan_fwd_f32", x, g, y, BC, HW)


def gate_chan_bwd_reduce(x, dy, g, dg, BC, HW):
    _chk(x, dy, g, dg)
    _call("cenet_gate_chan_bwd_reduce_f32", x, dy, g, dg, BC, HW)


def ccu_bwd_apply(x, dy, g, dz, u, amax, fc1, fc2, dfc1, dfc2, dx, B, Cn, HW, dx_add=None):
    _chk(x, dy, g, dz, u, amax, fc1, fc2, dfc1, dfc2, dx, dx_add)
    if dx_add is None:
        _call("cenet_ccu_bwd_apply_acc_f32", x, dy, g, dz, u, amax, fc1, fc2, dfc1, dfc2, dx, B, Cn, HW)
    else:
        _call("cenet_ccu_bwd_apply_add_acc_f32", x, dy, g, dz, u, amax, fc1, fc2, dfc1, dfc2, dx, dx_add, B, Cn, HW)


def srm_stats_fwd(x, u, amax, B, Cn, HW):
    _chk(x, u, amax)
    _call("cenet_srm_stats_fwd_f32", x, u, amax, B, Cn, HW)


def srm_conv_fwd(u, pwc, dwc, f, B, H, W):
    _chk(u, pwc, dwc, f)
    _call("cenet_srm_conv_fwd_f32", u, pwc, dwc, f, B, H, W)


def srm_conv_bwd(u, df, pwc, dwc, du, dpwc, ddwc, B, H, W):
    _chk(u, df, pwc, dwc, du, dpwc, ddwc)
    _call("cenet_srm_conv_bwd_acc_f32", u, df, pwc, dwc, du, dpwc, ddwc, B, H, W)


_NO_SRM_FUSED = __import__("os").environ.get("CENET_NO_SRM_FUSED") is not None  # measurement aid


def srm_fused_supported(B, H, W) -> bool:
    return (not _NO_CHANLOC) and (not _NO_SRM_FUSED) and bool(_lib.lib().cenet_srm_fused_supported(int(B), int(H), int(W)))


def srm_parts(B, H, W) -> int:
    return int(_lib.lib().cenet_srm_parts(int(B), int(H), int(W)))


def srm_conv_gelu_fwd(u, pwc, dwc, f, fa, part, B, H, W):
    _chk(u, pwc, dwc, f, fa, part)
    _call("cenet_srm_conv_gelu_fwd_f32", u, pwc, dwc, f, fa, part, B, H, W)


def gate_pix_bn_fwd(x, fa, part, G, fb, y, gamma, beta, eps, mean, var, rmean, rvar, momentum, nbt, B, Cn, HW):
    _chk(x, fa, part, fb, y, gamma, beta, mean, var, rmean, rvar)
    _call("cenet_gate_pix_bn_fwd_f32", x, fa, part, G, fb, y, gamma, beta, float(eps), mean, var, rmean, rvar, float(momentum), nbt, B,
          Cn, HW)


def srm_conv_bn_bwd(u, dfb, fa, f, mean, var, eps, gamma, pwc, dwc, part2, du, dpwc, ddwc, dgamma, dbeta, B, H, W):
    _chk(u, dfb, fa, f, mean, var, gamma, pwc, dwc, part2, du, dpwc, ddwc, dgamma, dbeta)
    _call("cenet_srm_conv_bn_bwd_acc_f32", u, dfb, fa, f, mean, var, float(eps), gamma, pwc, dwc, part2, du, dpwc, ddwc, dgamma,
          dbeta, B, H, W)


def gate_pix_fwd(x, f, y, B, Cn, HW):
    _chk(x, f, y)
    _call("cenet_gate_pix_fwd_f32", x, f, y, B, Cn, HW)


def gate_pix_bwd_reduce(x, dy, f, df, B, Cn, HW):
    _chk(x, dy, f, df)
    _call("cenet_gate_pix_bwd_reduce_f32", x, dy, f, df, B, Cn, HW)


def srm_bwd_apply(x, dy, f, u, du, amax, dx, B, Cn, HW):
    _chk(x, dy, f, u, du, amax, dx)
    _call("cenet_srm_bwd_apply_f32", x, dy, f, u, du, amax, dx, B, Cn, HW)


# ---- glue ---------------------------------------------------------------------------------------------------
def transpose(x, sxb, y, syb, B, R, Cc, x_off=0, y_off=0, add=None):
    """y[b] = x[b]^T (+ add[b], laid out like y with batch stride syb)"""
    _chk(x, y, add)
    if add is None:
        _call("cenet_transpose_f32", Ptr(x, x_off), L(sxb), Ptr(y, y_off), L(syb), B, R, Cc)
    else:
        _call("cenet_transpose_add_f32", Ptr(x, x_off), L(sxb), Ptr(y, y_off), L(syb), add, B, R, Cc)


def copy_batched(x, sxb, y, syb, B, n, accumulate=False, x_off=0, y_off=0):
    _chk(x, y)
    _call("cenet_copy_batched_f32", Ptr(x, x_off), L(sxb), Ptr(y, y_off), L(syb), B, L(n), int(accumulate))


def cat_channels(parts, joined, B, HW, split=False):
    """joined [B, sum c, HW] <- parts [B, c_j, HW] (or the reverse with split=True); at most four parts, one launch"""
    assert 1 <= len(parts) <= 4
    _chk(joined, *parts)
    ps = list(parts) + [None] * (4 - len(parts))
    cs = [int(t.shape[1]) for t in parts] + [0] * (4 - len(parts))
    _call("cenet_cat_channels_f32", *ps, *cs, joined, B, L(HW), int(split))


def split_channels_add(parts, adds, joined, B, HW):
    """parts[j] [B, c_j, HW] = channel slice j of joined + adds[j] (None: the slice alone); at most four parts, one launch"""
    assert 1 <= len(parts) <= 4 and len(adds) == len(parts)
    _chk(joined, *parts, *adds)
    ps = list(parts) + [None] * (4 - len(parts))
    ads = list(adds) + [None] * (4 - len(adds))
    cs = [int(t.shape[1]) for t in parts] + [0] * (4 - len(parts))
    _call("cenet_split_channels_add_f32", *ps, *ads, *cs, joined, B, L(HW))


def im2col_tok(src, dst, B, H, W, C, K, stride, pad, inverse=False):
    _chk(src, dst)
    _call("cenet_im2col_tok_f32", src, dst, B, H, W, C, K, stride, pad, int(inverse))


def patch_tok(src, dst, B, Ho, Wo, C, S, inverse=False):
    """Space-to-depth of a token map (inverse: patch rows back to tokens); pvtv2.py:93-95 kernel == stride conv."""
    _chk(src, dst)
    _call("cenet_patch_tok_f32", src, dst, B, Ho, Wo, C, S, int(inverse))


def scale_batch(x, s, y, B, n):
    _chk(x, s, y)
    _call("cenet_scale_batch_f32", x, s, y, B, L(n))


def act_fwd(x, y, n, act, slope=0.0):
    _chk(x, y)
    _call("cenet_act_fwd_f32", x, y, L(n), ACT[act], float(slope))


def act_bwd(pre, dy, dx, n, act, slope=0.0):
    _chk(pre, dy, dx)
    _call("cenet_act_bwd_f32", pre, dy, dx, L(n), ACT[act], float(slope))


def silu_mul_fwd(a, b, y, n):
    _chk(a, b, y)
    _call("cenet_silu_mul_fwd_f32", a, b, y, L(n))


def silu_mul_bwd(a, b, dy, da, db, n):
    _chk(a, b, dy, da, db)
    _call("cenet_silu_mul_bwd_f32", a, b, dy, da, db, L(n))


def mix_fwd(x, p, w, z, n):
    _chk(x, p, w, z)
    _call("cenet_mix_fwd_f32", x, p, w, z, L(n))


def mix_bwd(x, p, w, dz, dx, dp, dw, n):
    _chk(x, p, w, dz, dx, dp, dw)
    _call("cenet_mix_bwd_acc_f32", x, p, w, dz, dx, dp, dw, L(n))


def scale_residual_fwd(x, y, ls, out, B, Cn, HW):
    _chk(x, y, ls, out)
    _call("cenet_scale_residual_fwd_f32", x, y, ls, out, B, Cn, HW)


def scale_chan(g, ls, out, B, Cn, HW):
    _chk(g, ls, out)
    _call("cenet_scale_chan_f32", g, ls, out, B, Cn, HW)


def chan_dot(a, sab, b, sbb, out, B, Cn, HW):
    _chk(a, b, out)
    _call("cenet_chan_dot_acc_f32", a, L(sab), b, L(sbb), out, B, Cn, HW)


def col_sum(a, out, R, Cn):
    _chk(a, out)
    _call("cenet_col_sum_acc_f32", a, out, L(R), Cn)


def add_act_fwd(a, b, out, n, act="none", slope=0.0):
    _chk(a, b, out)
    _call("cenet_add_act_fwd_f32", a, b, out, L(n), ACT[act], float(slope))


def lrelu_bwd_from_out(out, dy, dx, n, slope):
    _chk(out, dy, dx)
    _call("cenet_lrelu_bwd_from_out_f32", out, dy, dx, L(n), float(slope))


def dseb_combine_fwd(y, rs, n, w, diff, ycoef, z, B, Cn, HW):
    _chk(y, w, diff, z, *rs)
    r = list(rs) + [None] * (3 - len(rs))
    _call("cenet_dseb_combine_fwd_f32", y, r[0], r[1], r[2], n, w, diff, float(ycoef), z, B, Cn, HW)


def dseb_combine_bwd(y, rs, n, w, diff, ycoef, dz, dy, drs, ddiff, dw, B, Cn, HW):
    _chk(y, w, diff, dz, dy, ddiff, dw, *rs, *drs)
    r = list(rs) + [None] * (3 - len(rs))
    d = list(drs) + [None] * (3 - len(drs))
    _call("cenet_dseb_combine_bwd_acc_f32", y, r[0], r[1], r[2], n, w, diff, float(ycoef), dz, dy, d[0], d[1], d[2], ddiff, dw, B,
          Cn, HW)


def diffattn_lambda_fwd(q1, k1, q2, k2, lam0, lam3, hd):
    _chk(q1, k1, q2, k2, lam3)
    _call("cenet_diffattn_lambda_fwd_f32", q1, k1, q2, k2, float(lam0), lam3, hd)


def diffattn_lambda_bwd(q1, k1, q2, k2, lam3, dlam, dq1, dk1, dq2, dk2, hd):
    _chk(q1, k1, q2, k2, lam3, dlam, dq1, dk1, dq2, dk2)
    _call("cenet_diffattn_lambda_bwd_acc_f32", q1, k1, q2, k2, lam3, dlam, dq1, dk1, dq2, dk2, hd)


def diffattn_combine_fwd(U, lam3, out, B, H, N, dv, eps, post):
    _chk(U, lam3, out)
    _call("cenet_diffattn_combine_fwd_f32", U, lam3, out, B, H, N, dv, float(eps), float(post))


def diffattn_combine_bwd(U, lam3, dout, dU, dlam, B, H, N, dv, eps, post):
    _chk(U, lam3, dout, dU, dlam)
    _call("cenet_diffattn_combine_bwd_acc_f32", U, lam3, dout, dU, dlam, B, H, N, dv, float(eps), float(post))


# ---- loss / optimiser --------------------------------------------------------------------------------------
def dice_ce_fwd(logits, labels, acc, loss, B, K, HW, w_dice, w_ce):
    _chk(logits, labels, acc, loss)
    _call("cenet_dice_ce_fwd_f32", logits, labels, acc, loss, B, K, HW, float(w_dice), float(w_ce))


def dice_ce_bwd(logits, labels, acc, gout, dlogits, B, K, HW, w_dice, w_ce):
    _chk(logits, labels, acc, gout, dlogits)
    _call("cenet_dice_ce_bwd_f32", logits, labels, acc, gout, dlogits, B, K, HW, float(w_dice), float(w_ce))


def seg_loss_fwd(logits, labels, acc, loss, B, K, H, W, w_dice, w_ce, w_bd):
    _chk(logits, labels, acc, loss)
    _call("cenet_seg_loss_fwd_f32", logits, labels, acc, loss, B, K, H, W, float(w_dice), float(w_ce), float(w_bd))


def seg_loss_bwd(logits, labels, acc, gout, dlogits, B, K, H, W, w_dice, w_ce, w_bd):
    _chk(logits, labels, acc, gout, dlogits)
    _call("cenet_seg_loss_bwd_f32", logits, labels, acc, gout, dlogits, B, K, H, W, float(w_dice), float(w_ce), float(w_bd))


def argmax_counts(logits, labels, pred, counts, B, K, HW):
    _chk(logits, labels, pred)
    _call("cenet_argmax_counts_f32", logits, labels, pred, counts, B, K, HW)


def surface_border(mask_u8, border_u8, D, H, W):
    _chk(mask_u8, border_u8)
    _call("cenet_surface_border_u8", mask_u8, border_u8, D, H, W)


def min_sqdist(a_i32, b_i32, out_i32):
    _chk(a_i32, b_i32, out_i32)
    _call("cenet_min_sqdist_i32", a_i32, a_i32.shape[0], b_i32, b_i32.shape[0], out_i32)


def sgd_step(p, g, buf, hyper5, n, shadow=None):
    """shadow: bf16 copy of the parameters, rewritten by the same kernel (the GEMM operand of the throughput mode)"""
    _chk(p, g, buf, hyper5, shadow)
    if shadow is None:
        _call("cenet_sgd_step_f32", p, g, buf, hyper5, L(n))
    else:
        rc = _lib.lib().cenet_sgd_step_shadow_f32(P(p), P(g), P(buf), P(hyper5), L(n), P(shadow), stream())
        _lib.check(rc, "cenet_sgd_step_shadow_f32")


def cast(x: torch.Tensor, dtype) -> torch.Tensor:
    """fp32 <-> bf16 copy of a contiguous tensor (cenet_cast_*); returns x itself when it already has `dtype`"""
    if x.dtype == dtype:
        return x
    _chk(x)
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=dtype)
    name = "cenet_cast_f32_to_bf16" if dtype == BF16 else "cenet_cast_bf16_to_f32"
    _lib.check(getattr(_lib.lib(), name)(P(x), P(y), L(x.numel()), stream()), name)
    return y


def cast_clear(x: torch.Tensor, y: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """y (bf16) = x (fp32, contiguous) [+ bias over the last axis]; x = 0 afterwards (see ops._ZeroWs)"""
    _chk(x, y, bias)
    assert x.dtype == torch.float32 and y.dtype == BF16 and x.numel() == y.numel()
    if bias is None:
        _lib.check(_lib.lib().cenet_cast_clear_f32_to_bf16(P(x), P(y), L(x.numel()), stream()), "cenet_cast_clear_f32_to_bf16")
    else:
        _lib.check(_lib.lib().cenet_cast_clear_bias_f32_to_bf16(P(x), P(y), P(bias), C.c_int(bias.numel()), L(x.numel()), stream()),
                   "cenet_cast_clear_bias_f32_to_bf16")


def cast_into(x: torch.Tensor, y: torch.Tensor):
    name = "cenet_cast_f32_to_bf16" if y.dtype == BF16 else "cenet_cast_bf16_to_f32"
    assert x.dtype != y.dtype and x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous()
    _lib.check(getattr(_lib.lib(), name)(P(x), P(y), L(x.numel()), stream()), name)


def wq(W: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """The weight as the GEMM operand for activations of `like`'s type: W itself (fp32) or its bf16 shadow.
    A parameter that lives in a `ParamArena` has a slot in the arena's bf16 buffer, refreshed by the fused SGD kernel; any
    other tensor gets a private shadow.  A shadow is re-cast whenever the tensor was modified in place through torch since
    (`Tensor._version`), e.g. by load_state_dict or a torch optimizer."""
    if like.dtype != BF16 or W is None:
        return W
    sh = getattr(W, "_cenet_shadow", None)
    if sh is None:
        slot = getattr(W, "_cenet_arena_slot", None)
        if slot is not None and slot[0].params.device == W.device:
            slot[0].enable_shadow()
            sh = W._cenet_shadow
    if sh is None or sh.shape != W.shape or sh.device != W.device:
        sh = torch.empty(W.shape, device=W.device, dtype=BF16)
        W._cenet_shadow = sh
        W._cenet_shadow_ver = -1
    if W._cenet_shadow_ver != W._version:
        cast_into(W.detach() if W.is_contiguous() else W.detach().contiguous(), sh)
        W._cenet_shadow_ver = W._version
    return sh


def zero_(t: torch.Tensor):
    """zero-fill with the library's own kernel (a kernel node replays faithfully in a hipGraph; hipMemsetAsync did not)"""
    _chk(t)
    nbytes = t.numel() * t.element_size()
    assert t.is_contiguous(), "zero_: contiguous tensor"
    if nbytes == 0:
        return t
    if nbytes % 4 == 0 and t.data_ptr() % 4 == 0:
        _lib.check(_lib.lib().cenet_zero_f32(P(t), L(nbytes // 4), stream()), "cenet_zero_f32")
    else:  # a bf16 tensor with an odd element count / an odd-element view: head and tail bytes handled by the kernel
        _lib.check(_lib.lib().cenet_zero_bytes(P(t), L(nbytes), stream()), "cenet_zero_bytes")
    return t


def conv_direct_supported(Cin: int, Cout: int, k: int, stride: int, pad: int) -> bool:
    return bool(_lib.lib().cenet_conv_direct_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def conv_direct(x, w, y, B, Cin, Cout, H, W, k, dgrad):
    """direct convolution on bf16 tensors (LDS halo tiles, weights resident in LDS; w = fp32 master weight);
    dgrad: x=dY, w=forward weight, y=dX."""
    _chk(x, w, y)
    assert x.dtype == BF16 and y.dtype == BF16 and w.dtype == torch.float32
    rc = _lib.lib().cenet_conv_direct_bf16(P(x), P(w), P(y), B, Cin, Cout, H, W, k, int(dgrad), stream())
    _lib.check(rc, "cenet_conv_direct_bf16")


def conv_c1_supported(Cin: int, Cout: int, k: int, stride: int, pad: int) -> bool:
    return bool(_lib.lib().cenet_conv_c1_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def conv_c1_fwd(x, w, y, B, Cout, H, W, k):
    """one-channel bf16 image -> Cout <= 32 channels, stride 1, same padding (conv_c1.hip)"""
    _chk(x, w, y)
    assert x.dtype == BF16 and y.dtype == BF16 and w.dtype == torch.float32
    _lib.check(_lib.lib().cenet_conv_c1_fwd_bf16(P(x), P(w), P(y), B, Cout, H, W, k, stream()), "cenet_conv_c1_fwd_bf16")


def conv_c1_wgrad(x, dy, dw, B, Cout, H, W, k):
    _chk(x, dy, dw)
    assert x.dtype == BF16 and dy.dtype == BF16 and dw.dtype == torch.float32
    _lib.check(_lib.lib().cenet_conv_c1_wgrad_bf16(P(x), P(dy), P(dw), B, Cout, H, W, k, stream()), "cenet_conv_c1_wgrad_bf16")


def pw_fewout_supported(Cin: int, Cout: int) -> bool:
    return bool(_lib.lib().cenet_pw_fewout_supported(int(Cin), int(Cout)))


def pw_fewout_wgrad_supported(Cin: int, Cout: int) -> bool:
    return bool(_lib.lib().cenet_pw_fewout_wgrad_supported(int(Cin), int(Cout)))


def pw_fewout_fwd(x, W, bias, y, B, Cin, Cout, HW):
    """64 -> few channels 1x1 conv with bias on bf16 tensors (conv_c1.hip); W: bf16 [Cout, 64]"""
    _chk(x, W, bias, y)
    assert x.dtype == BF16 and y.dtype == BF16 and W.dtype == BF16
    _lib.check(_lib.lib().cenet_pw_fewout_fwd_bf16(P(x), P(W), P(bias) if bias is not None else None, P(y), B, Cin, Cout, L(HW),
                                                   stream()), "cenet_pw_fewout_fwd_bf16")


def pw_fewout_dgrad(dy, W, dx, B, Cin, Cout, HW):
    _chk(dy, W, dx)
    assert dy.dtype == BF16 and dx.dtype == BF16 and W.dtype == BF16
    _lib.check(_lib.lib().cenet_pw_fewout_dgrad_bf16(P(dy), P(W), P(dx), B, Cin, Cout, L(HW), stream()), "cenet_pw_fewout_dgrad_bf16")


def pw_fewout_wgrad(x, dy, dW, dbias, B, Cin, Cout, HW):
    _chk(x, dy, dW, dbias)
    assert x.dtype == BF16 and dy.dtype == BF16 and dW.dtype == torch.float32
    _lib.check(_lib.lib().cenet_pw_fewout_wgrad_bf16(P(x), P(dy), P(dW), P(dbias) if dbias is not None else None, B, Cin, Cout,
                                                     L(HW), stream()), "cenet_pw_fewout_wgrad_bf16")


def pw_small_supported(G: int) -> bool:
    return bool(_lib.lib().cenet_pw_small_supported(int(G)))


def pw_small(x, W, y, B, groups, G, HW, transpose=False):
    """groups x (G x G) bias-free 1x1 convs on bf16 NCHW tensors, G <= 40 (conv_c1.hip); W: bf16 [groups, G, G]"""
    _chk(x, W, y)
    assert x.dtype == BF16 and y.dtype == BF16 and W.dtype == BF16
    _lib.check(_lib.lib().cenet_pw_small_bf16(P(x), P(W), P(y), B, groups, G, L(HW), int(transpose), stream()),
               "cenet_pw_small_bf16")


def conv_wgrad_direct_supported(Cin: int, Cout: int, k: int, stride: int, pad: int) -> bool:
    return bool(_lib.lib().cenet_conv_wgrad_direct_supported(int(Cin), int(Cout), int(k), int(stride), int(pad)))


def conv_wgrad_direct(x, dy, dw, B, Cin, Cout, H, W, k):
    """bf16-operand direct weight gradient (LDS tiles of dY and the X halo, funnel-shifted tap windows): dw += dY (*) x."""
    _chk(x, dy, dw)
    assert x.dtype == BF16 and dy.dtype == BF16 and dw.dtype == torch.float32
    fn = _lib.lib().cenet_conv_wgrad_direct_ws_floats
    fn.restype = C.c_long
    ws = torch.empty(int(fn(int(Cin), int(Cout), int(k))), device=x.device, dtype=torch.float32)
    rc = _lib.lib().cenet_conv_wgrad_direct_bf16(P(x), P(dy), P(dw), P(ws), B, Cin, Cout, H, W, k, stream())
    _lib.check(rc, "cenet_conv_wgrad_direct_bf16")


# ---- precision mode ------------------------------------------------------------------------------------------------
# The element type of the tensors decides which kernels run: fp32 tensors take the exact fp32 MFMA chain (parity mode),
# bf16 tensors the throughput path (bf16 in HBM and in the MFMA operands, fp32 accumulation / softmax / statistics).  This
# flag only tells `CENet.forward` to cast its input (and thereby the whole network) to bf16.
_COMPUTE_BF16 = False


def set_compute_bf16(on: bool) -> bool:
    """False (default): CENet runs in fp32, the mode every parity claim is made in.  True: CENet casts its input to bf16 and
    all activations, activation gradients and GEMM operands are bf16 (weights through a bf16 shadow of the fp32 master
    copy); logits come back as bf16.  Returns the previous setting."""
    global _COMPUTE_BF16
    old = _COMPUTE_BF16
    _COMPUTE_BF16 = bool(on)
    return old


def get_compute_bf16() -> bool:
    return _COMPUTE_BF16


def augment_acdc(pool_img, pool_lab, tab, dp, stage_img, stage_lab, stage_stride, max_h, max_w, out_img, out_lab, B, OH, OW):
    """cenet_augment_acdc (augment.hip): gather + augment one training batch from the device-resident slices"""
    assert pool_img.dtype == torch.float32 and pool_lab.dtype == torch.uint8 and tab.dtype == torch.int64 and dp.dtype == torch.float64
    assert stage_img.dtype == torch.float64 and stage_lab.dtype == torch.uint8 and out_img.dtype == torch.float32
    for t in (pool_img, pool_lab, tab, dp, stage_img, stage_lab, out_img, out_lab):
        assert t.is_contiguous()
    rc = _lib.lib().cenet_augment_acdc(P(pool_img), P(pool_lab), P(tab), P(dp), P(stage_img), P(stage_lab), L(stage_stride),
                                       C.c_int(max_h), C.c_int(max_w), P(out_img), P(out_lab), C.c_int(B), C.c_int(OH), C.c_int(OW),
                                       stream())
    _lib.check(rc, "cenet_augment_acdc")


def layernorm_fwd_acc(acc, bias, xpre, gamma, beta, y, mean, rstd, rows, Cn, eps):
    """cenet_layernorm_fwd_acc_bf16: LayerNorm straight from a split-K fp32 accumulator (+ bias); acc is zero afterwards"""
    _chk(acc, bias, xpre, gamma, beta, y, mean, rstd)
    assert acc.dtype == torch.float32 and xpre.dtype == BF16 and y.dtype == BF16
    rc = _lib.lib().cenet_layernorm_fwd_acc_bf16(P(acc), P(bias), P(xpre), P(gamma), P(beta), P(y), P(mean), P(rstd), C.c_int(rows),
                                                 C.c_int(Cn), C.c_float(eps), stream())
    _lib.check(rc, "cenet_layernorm_fwd_acc_bf16")
