"""Flat parameter/gradient arena, fused SGD and the poly LR schedule.

Mirrors the reference's optimiser plumbing (src/utils/core.py:12-41: torch.optim.SGD(lr, momentum 0.9, weight_decay)
and LambdaLR (1 - it/max)^0.9) on top of ONE contiguous fp32 buffer for parameters, one for gradients and one for
momentum, so that per step there is one memset, one fused update kernel and a handful of large all-reduces instead
of 630 small ones.  `torch.optim.SGD(net.parameters())` keeps working on arena-backed parameters
(their `.data` / `.grad` are views), so the reference's `get_optimizer` can also be used unchanged.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import kern

ALIGN = 64  # elements (256 B)


class ParamArena:
    """Re-homes every trainable parameter of `model` into one flat buffer (and its gradient into another).

    segments: optional ordered list of (name, predicate(param_name) -> bool); parameters are laid out segment by
    segment so that a gradient segment is one contiguous slice (used as an all-reduce bucket). Call AFTER the model
    has been moved to its device.
    """

    def __init__(self, model: nn.Module, segments: Optional[Sequence[Tuple[str, Callable[[str], bool]]]] = None):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        dev = named[0][1].device
        if segments is None:
            segments = [("all", lambda n: True)]
        # parameter groups that a module wants back to back WITHOUT padding between the members (`arena_groups()` on any
        # sub-module: lists of equally shaped parameters, e.g. the same weight of the three dilated branches of a
        # MultiOrderDWConv), so that one kernel launch can address them as one [n, ...] tensor (cenet_amd.ops.merged_param)
        name_of = {id(p): n for n, p in named}
        group_of = {}
        for m in model.modules():
            fn = getattr(m, "arena_groups", None)
            if fn is None:
                continue
            for grp in fn():
                names = [name_of[id(q)] for q in grp if id(q) in name_of]
                if len(names) == len(grp) and len(names) > 1:
                    for n in names:
                        group_of[n] = names
        byname = dict(named)
        order, self.segments = [], []
        taken = set()
        off = 0
        for sname, pred in segments:
            start = off
            for n, p in named:
                if n in taken or not pred(n):
                    continue
                members = group_of.get(n, [n])
                if any(not pred(k) or k in taken for k in members):
                    members = [n]  # a group never straddles a segment
                for k in members:
                    taken.add(k)
                    order.append((k, byname[k], off))
                    off += byname[k].numel()
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            self.segments.append((sname, start, off))
        left = [n for n, _ in named if n not in taken]
        if left:
            raise ValueError(f"parameters not covered by any segment: {left[:5]}")
        self.numel = off
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        self.index = {}
        with torch.no_grad():
            for n, p, o in order:
                view = self.params[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grads[o:o + p.numel()].view(p.shape)
                p._cenet_grad_home = self._home(o, p.numel(), tuple(p.shape))  # ops.grad_buf: re-attach after .grad = None
                self.index[n] = (o, p.numel())
        self._plist = [p for _, p, _ in order]
        self._offs = [o for _, _, o in order]
        # bf16 shadow of the parameters (the GEMM operand of the throughput mode), created on first use by kern.wq
        self.shadow = None
        for p, o in zip(self._plist, self._offs):
            p._cenet_arena_slot = (self, o)

    def enable_shadow(self):
        """One flat bf16 copy of `params`; every parameter's `_cenet_shadow` is a view into it.  `FusedSGD.step` rewrites it
        in the same kernel that updates the fp32 master copy."""
        if self.shadow is not None:
            return
        self.shadow = torch.empty(self.numel, device=self.params.device, dtype=torch.bfloat16)
        kern.cast_into(self.params, self.shadow)
        for p, o in zip(self._plist, self._offs):
            p._cenet_shadow = self.shadow[o:o + p.numel()].view(p.shape)
            p._cenet_shadow_ver = p._version

    def refresh_shadow(self):
        """Re-cast the whole bf16 shadow from the fp32 master copy.  Needed after any RAW write to `self.params` (the flat
        broadcast of `GradReducer.broadcast_state`, a checkpoint restored into the arena): `kern.wq` detects stale shadows by
        each parameter's own `_version`, which a write through the flat buffer does not bump."""
        if self.shadow is None:
            return
        kern.cast_into(self.params, self.shadow)
        for p in self._plist:
            p._cenet_shadow_ver = p._version

    def _home(self, o: int, n: int, shape):
        def home():
            g = self.grads[o:o + n]
            kern.zero_(g)  # the slot still holds the previous step's gradient
            return g.view(shape)
        return home

    def zero_grad(self):
        """One memset for all gradients (re-attaches views if a caller set .grad to None)."""
        from . import ops
        ops.wgrad_join()
        kern.zero_(self.grads)
        for p, o in zip(self._plist, self._offs):
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * o:
                p.grad = self.grads[o:o + p.numel()].view(p.shape)

    def segment_grad(self, i: int) -> torch.Tensor:
        _, s, e = self.segments[i]
        return self.grads[s:e]


class FusedSGD:
    """torch.optim.SGD(momentum, weight_decay, dampening 0, nesterov False) semantics, one kernel launch per step."""

    def __init__(self, arena: ParamArena, lr: float, momentum: float = 0.9, weight_decay: float = 1e-4,
                 grad_scale: float = 1.0):
        self.arena = arena
        # ONE persistent group dict, as torch.optim exposes it: `for g in opt.param_groups: g["lr"] = x` (warm-up code, torch
        # LR schedulers) must take effect.  lr / momentum / weight_decay are read from it at every step.
        self.param_groups = [{"lr": lr, "momentum": momentum, "weight_decay": weight_decay, "params": arena._plist}]
        self.grad_scale = grad_scale
        self.buf = torch.zeros_like(arena.params)
        self.hyper = torch.zeros(5, device=arena.params.device, dtype=torch.float32)
        self._steps = 0
        self._sync_hyper()

    # lr / momentum / weight_decay live in param_groups[0] (torch.optim idiom); these properties are the short spelling
    lr = property(lambda self: self.param_groups[0]["lr"], lambda self, v: self.param_groups[0].__setitem__("lr", v))
    momentum = property(lambda self: self.param_groups[0]["momentum"],
                        lambda self, v: self.param_groups[0].__setitem__("momentum", v))
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"],
                            lambda self, v: self.param_groups[0].__setitem__("weight_decay", v))

    def _sync_hyper(self):
        vals = [self.lr, self.momentum, self.weight_decay, self.grad_scale, 1.0 if self._steps == 0 else 0.0]
        if self.hyper.is_cuda:
            # asynchronous upload from PINNED staging (a pageable temporary could be recycled by the host before the copy
            # executes when the host runs ahead of the GPU, e.g. under hipGraph replay).  4 slots rotate; a slot is reused
            # only after the copy that last read it has executed (its event), which also bounds how far the host can run
            # ahead of the GPU: 4 steps.
            if not hasattr(self, "_stage"):
                self._stage = [torch.zeros(5, dtype=torch.float32).pin_memory() for _ in range(4)]
                self._stage_ev = [None] * 4
                self._slot = 0
            i = self._slot
            self._slot = (i + 1) % len(self._stage)
            if self._stage_ev[i] is not None:
                self._stage_ev[i].synchronize()
            h = self._stage[i]
            for j, v in enumerate(vals):
                h[j] = v
            self.hyper.copy_(h, non_blocking=True)
            if self._stage_ev[i] is None:
                self._stage_ev[i] = torch.cuda.Event()
            self._stage_ev[i].record()
        else:
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32))

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    def set_lr(self, lr: float):
        self.lr = lr

    def step(self, sync_hyper: bool = True):
        """sync_hyper=False: the caller has uploaded the hyper-parameters already (`prepare()`); needed when the step is
        captured in a HIP graph, where a host->device copy is not allowed inside the captured region."""
        if sync_hyper:
            self._sync_hyper()
        from . import ops
        ops.wgrad_join()  # weight gradients issued on the side stream (ops._wgrad_side) must have landed
        kern.sgd_step(self.arena.params, self.arena.grads, self.buf, self.hyper, self.arena.numel, self.arena.shadow)
        self._steps += 1

    def prepare(self):
        """Upload lr / momentum / wd / grad_scale / first-step flag for the NEXT step (call before a graph replay)."""
        self._sync_hyper()

    def state_dict(self):
        """momentum per parameter NAME (`momentum`: name -> tensor): the flat arena layout (segments, arena groups, padding) is
        an implementation detail that may differ between the run that saved and the run that loads"""
        idx = self.arena.index
        return {"momentum": {n: self.buf[o:o + k].detach().clone() for n, (o, k) in idx.items()}, "steps": self._steps,
                "lr": self.lr}

    def load_state_dict(self, sd):
        if "momentum" in sd:
            idx = self.arena.index
            missing = [n for n in idx if n not in sd["momentum"]]
            if missing:
                raise KeyError(f"momentum missing for parameters {missing[:5]}")
            for n, (o, k) in idx.items():
                self.buf[o:o + k].copy_(sd["momentum"][n].reshape(-1).to(self.buf.device))
        else:
            # (the flat 'buf' of files written before the arena grew groups: the layout is not recorded in the file, and a
            # same-size buffer of another layout would be silently misassigned)
            raise ValueError("training state without per-parameter momentum ('momentum': name -> tensor); re-save it")
        self._steps, self.lr = sd["steps"], sd["lr"]


class PolyLR:
    """core.py:31: lr = base * (1 - step / max_iterations) ** 0.9 (LambdaLR semantics: step() after optimizer.step())."""

    def __init__(self, optimizer: FusedSGD, max_iterations: int, power: float = 0.9):
        self.opt, self.max_it, self.power = optimizer, max_iterations, power
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def get_last_lr(self) -> List[float]:
        return [self.opt.lr]

    def step(self):
        self.last_epoch += 1
        self.opt.set_lr(self.base_lr * (1 - self.last_epoch / self.max_it) ** self.power)


def cenet_segments():
    """Arena layout for CENet in reverse-forward order: the gradient of each segment is complete (and can start its
    all-reduce) when backward reaches the segment's input (cenet_amd.parallel)."""
    return [("head+decoder", lambda n: n.startswith(("out.", "decoder."))),
            ("stage4", lambda n: n.startswith(("backbone.patch_embed4", "backbone.block4", "backbone.norm4"))),
            ("stage3", lambda n: n.startswith(("backbone.patch_embed3", "backbone.block3", "backbone.norm3"))),
            ("stage2", lambda n: n.startswith(("backbone.patch_embed2", "backbone.block2", "backbone.norm2"))),
            ("stage1", lambda n: n.startswith(("backbone.patch_embed1", "backbone.block1", "backbone.norm1")))]
