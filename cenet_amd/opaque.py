"""`CENet.forward` as ONE opaque `torch.library` operator for callers that TRACE the model instead of running it:
`torch.compile(net, mode='default', fullgraph=True)` (reference main_acdc.py:188-191), `torch.jit.trace` (fvcore's
FlopCountAnalysis behind utils/utils.py:171-181, main_acdc.py:128) and `torch.export`.

The network's operators are ctypes launches of libcenet_hip.so wrapped in autograd.Functions: a tracer can neither look inside
them nor needs to.  `cenet_amd::forward(x, anchor, ...)` runs the eager forward (building the network's own autograd graph when
gradients are wanted) and is registered with a fake (shape) implementation and an autograd formula, so the traced graph
contains one node with the right output shape, and its backward one `cenet_amd::backward` node, which differentiates the stored
graph — parameter gradients are accumulated into `.grad` in place by the kernels, exactly as in eager mode.  `anchor` is one of
the network's parameters: it makes the output require a gradient in the tracer's eyes (its own returned gradient is zero).
Nothing here touches the eager path."""
from __future__ import annotations

import weakref

import torch

_NETS = {}
_next_handle = [1]


def register(net) -> int:
    """called EAGERLY (CENet.__init__ / __setstate__, i.e. also for deepcopies and unpickled models): a tracer must find the
    handle as a plain int attribute, it cannot be handed out while tracing"""
    h = _next_handle[0]
    _next_handle[0] += 1
    net.__dict__["_cenet_handle"] = h
    _NETS[h] = weakref.ref(net, lambda _, h=h: _NETS.pop(h, None))
    return h


class _autograd_keys_included:
    """the autograd dispatch keys back in force inside a `with` block (an operator's backend implementation runs with the
    AutogradFunctionality key excluded in thread-local state)"""

    def __enter__(self):
        K = torch._C.DispatchKey
        excl = torch._C._dispatch_tls_local_exclude_set()
        for k in (K.AutogradFunctionality, K.AutogradOther, K.AutogradNestedTensor):
            excl = excl.remove(k)
        self.guard = torch._C._ForceDispatchKeyGuard(torch._C._dispatch_tls_local_include_set(), excl)
        self.guard.__enter__()

    def __exit__(self, *exc):
        self.guard.__exit__(*exc)
        return False


def _net(handle: int):
    net = _NETS[handle]()
    if net is None:
        raise RuntimeError("cenet_amd::forward: the CENet module this graph was traced from no longer exists")
    return net


_MAX_LIVE = 8  # forwards of one network whose backward has not run yet (more: the oldest graph is dropped)
_next_token = [1]


@torch.library.custom_op("cenet_amd::forward", mutates_args=())
def forward_op(x: torch.Tensor, anchor: torch.Tensor, handle: int, num_classes: int, bf16: bool,
               grad: bool) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (logits, token).  The token (a host int64 scalar) names the autograd graph this call built; cenet_amd::backward looks
    its graph up BY that token, so forwards and backwards pair correctly in any order (two forwards then the later one's
    backward first; a forward whose backward never runs)."""
    net = _net(handle)
    token = 0
    if grad:
        # An operator's backend implementation runs BELOW the autograd dispatch keys (they are excluded in thread-local state):
        # plain aten views / reshapes between the network's autograd.Functions would silently stop carrying requires_grad and
        # cut whole branches out of the network's own graph.  Re-include them for the duration of the forward.
        with _autograd_keys_included(), torch.enable_grad():
            out = net._forward(x)
        live = net.__dict__.setdefault("_cenet_live", {})  # token -> the network's own autograd graph (its output)
        token = _next_token[0]
        _next_token[0] += 1
        live[token] = out
        while len(live) > _MAX_LIVE:  # graphs nobody differentiated (skipped step, train-mode forward used for metrics only)
            live.pop(next(iter(live)))
    else:
        with torch.no_grad():
            out = net._forward(x)
    return out.detach(), torch.tensor([token], dtype=torch.int64)


@forward_op.register_fake
def _(x, anchor, handle, num_classes, bf16, grad):
    return (x.new_empty((x.shape[0], num_classes, x.shape[2], x.shape[3]), dtype=torch.bfloat16 if bf16 else torch.float32),
            torch.empty((1,), dtype=torch.int64, device="cpu"))


@torch.library.custom_op("cenet_amd::backward", mutates_args=())
def backward_op(g: torch.Tensor, token: torch.Tensor, handle: int) -> torch.Tensor:
    net = _net(handle)
    live = net.__dict__.get("_cenet_live") or {}
    out = live.pop(int(token.item()), None)  # (host tensor: no device sync)
    if out is None:
        raise RuntimeError("cenet_amd::backward: the graph of this forward is gone (its backward already ran, the forward ran "
                           f"without gradients, or more than {_MAX_LIVE} forwards of the network were left undifferentiated)")
    out.backward(g.to(out.dtype).contiguous())
    return torch.zeros(1, device=g.device, dtype=torch.float32)


@backward_op.register_fake
def _(g, token, handle):
    return g.new_zeros((1,), dtype=torch.float32)


def _setup_context(ctx, inputs, output):
    ctx.handle = inputs[2]
    ctx.anchor_shape = inputs[1].shape
    ctx.save_for_backward(output[1])


def _backward(ctx, g, g_token=None):
    (token,) = ctx.saved_tensors
    z = torch.ops.cenet_amd.backward(g, token, ctx.handle)
    return None, z.sum().to(g.dtype).expand(ctx.anchor_shape) * 0.0, None, None, None, None


torch.library.register_autograd("cenet_amd::forward", _backward, setup_context=_setup_context)


def forward(net, x: torch.Tensor, bf16: bool) -> torch.Tensor:
    """what CENet.forward returns under a tracer: one opaque node"""
    anchor = net.out.w
    grad = bool(net.training and torch.is_grad_enabled())
    return torch.ops.cenet_amd.forward(x, anchor, net._cenet_handle, net.out.out[1].conv.conv.out_channels, bf16, grad)[0]
