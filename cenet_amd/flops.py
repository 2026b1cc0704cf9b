"""FLOP count of `CENet.forward` for the reference's `print_param_flops` (utils/utils.py:171-181, main_acdc.py:128), which runs
fvcore's FlopCountAnalysis = a `torch.jit.trace` plus a per-operator table.  Under a tracer the forward is ONE opaque operator
(`cenet_amd::forward`, cenet_amd/opaque.py); this module supplies its table entry.

`count(net, shape)` DRY-RUNS an eval forward of the fp32 path on host tensors with the kernel library replaced by a recorder:
no kernel is launched, the Python side allocates its outputs as always, and every C-ABI call is priced from its arguments by
fvcore's published rules (one multiply-accumulate = 1 flop; conv / linear / matmul: output elements x reduction length;
layer_norm: 5 per element; batch_norm in eval mode: 1 per element; bilinear upsampling: 4 per output element; adaptive average
pooling: 1 per input element; everything else 0).  fvcore itself is absent from this image, so the rule restatement is
unpinned; the reference's own printout for the ACDC preset (12.76 G) is the anchor the test holds it to."""
from __future__ import annotations

import collections
import ctypes as C

import torch

from . import _lib


def _v(a):
    return a.value if hasattr(a, "value") else a


class _Recorder:
    """stands in for libcenet_hip.so: every entry point returns 0 and is priced by name"""

    def __init__(self):
        self.flops = collections.Counter()

    def __getattr__(self, name):
        def fn(*args):
            self._price(name, args)
            return 0
        fn.restype = None
        return fn

    def _price(self, name, a):
        n = name.replace("cenet_", "")
        for suf in ("_f32", "_bf16"):
            if n.endswith(suf):
                n = n[:-len(suf)]
        f = 0
        if n == "gemm":
            M, N, K, nbatch, _, nkb = (_v(x) for x in a[3:9])
            f = M * N * K * nbatch * nkb
        elif n == "flash_attn_fwd":
            t = a[0]._obj
            f = t.B * t.H * t.Nq * t.Nk * (t.D + t.Dv)
        elif n == "dwconv3x3_tok":
            B, Cn, H, W = (_v(x) for x in a[5:9])
            f = B * Cn * H * W * 9
        elif n == "dwconv3x3_nchw":
            B, Cn, H, W = (_v(x) for x in a[8:12])
            f = B * Cn * H * W * 9
        elif n == "layernorm_fwd":
            f = 5 * _v(a[6]) * _v(a[7])
        elif n == "bn_apply":
            B, Cn, HW = (_v(x) for x in a[11:14])
            f = B * Cn * HW
        elif n == "bilinear_fwd":
            B, Cn, _, _, Ho, Wo = (_v(x) for x in a[4:10])
            f = 4 * B * Cn * Ho * Wo
        elif n == "adaptive_avgpool_fwd":
            B, Cn, Hi, Wi = (_v(x) for x in a[4:8])
            f = B * Cn * Hi * Wi
        if f:
            self.flops[n] += int(f)


def count(net, shape, by_op: bool = False):
    """fvcore-rule FLOPs of one eval forward of `net` on an input of `shape` ([B, C, H, W]); by_op: the per-operator Counter"""
    rec = _Recorder()
    old = (_lib._LIB, _lib._HOSTSIM)
    orig, was_training = net, net.training
    dev = next(net.parameters()).device
    _lib._LIB, _lib._HOSTSIM = rec, True  # (host tensors pass the device check; nothing is launched)
    try:
        net.eval()
        if dev.type != "cpu":
            import copy
            net = copy.deepcopy(net).to("cpu")  # (BatchNorm running statistics etc. are only read)
        # differential attention: the kernels multiply EACH of the 2H softmax maps with V (no N x N map is ever stored), the
        # reference subtracts the two maps of a head first and multiplies once (multihead_diffattn.py:112-116).  The count
        # follows the reference's operator list: per module, H N^2 2hd = N^2 E multiply-accumulates fewer.
        from .networks.cenet.modules.multihead_diffattn import MultiheadDiffAttn
        hooks = [m.register_forward_pre_hook(lambda mod, inp: rec.flops.update(
            {"diffattn_maps_combined_first": -int(inp[0].shape[0] * inp[0].shape[1] ** 2 * mod.embed_dim)}))
            for m in net.modules() if isinstance(m, MultiheadDiffAttn)]
        try:
            with torch.no_grad():
                net._forward(torch.zeros(tuple(shape), dtype=torch.float32))
        finally:
            for h in hooks:
                h.remove()
    finally:
        _lib._LIB, _lib._HOSTSIM = old
        orig.train(was_training)  # (the CALLER's module: `net` may be the host copy by now)
    return rec.flops if by_op else sum(rec.flops.values())


def _fvcore_handle(inputs, outputs):
    """fvcore JitModelAnalysis handle for `cenet_amd::forward` (inputs / outputs are torch._C.Value lists)"""
    from . import opaque
    shape = inputs[0].type().sizes()
    net = opaque._net(inputs[2].toIValue())
    return collections.Counter({"cenet_amd::forward": count(net, shape)})


def register_with_fvcore() -> bool:
    """puts the handle into fvcore's default table, so that an unmodified `FlopCountAnalysis(net, x).total()` counts the opaque
    operator; returns False when fvcore is not installed"""
    try:
        import fvcore.nn.flop_count as fc
    except Exception:
        return False
    table = getattr(fc, "_DEFAULT_SUPPORTED_OPS", None)
    if table is None:
        return False
    table["cenet_amd::forward"] = _fvcore_handle
    return True
