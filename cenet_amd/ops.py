"""Autograd operators of the CENet hot path, forward AND backward on hand-written HIP kernels.

Each class is a `torch.autograd.Function` whose two methods only allocate tensors (torch caching allocator =
"plumbing") and launch kernels from libcenet_hip.so through `cenet_amd.kern`.  There is no PyTorch-op
fallback: on a non-GPU tensor `kern` raises.

Parameter gradients are ACCUMULATED IN PLACE into `param.grad` (created zero-filled if absent) and the
Function returns `None` for them; this lets all gradients live in one flat arena (cenet_amd.optim) that is
zeroed with a single memset, updated by one fused SGD launch and all-reduced in large buckets.
Reference file:line citations are relative to /root/reference/src/.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Optional, Sequence

import torch
from torch.autograd import Function

from . import kern

Tensor = torch.Tensor


def _empty(shape, ref: Tensor, dtype=torch.float32) -> Tensor:
    """fp32 (statistics, workspaces, parameter-gradient scratch) unless a dtype is given"""
    return torch.empty(shape, device=ref.device, dtype=dtype)


def _act(shape, ref: Tensor) -> Tensor:
    """an ACTIVATION tensor: the storage type of `ref` (fp32 in parity mode, bf16 in throughput mode)"""
    return torch.empty(shape, device=ref.device, dtype=ref.dtype)


def _bf(t: Tensor) -> bool:
    return t.dtype == torch.bfloat16


def _acc32(shape, ref: Tensor, like: Optional[Tensor] = None) -> Tensor:
    """fp32 accumulator for an atomic epilogue (zero-filled, or a copy of `like`)"""
    if like is not None:
        return kern.cast(like, torch.float32) if like.dtype != torch.float32 else like.clone()
    return _zeros(shape, ref)


def _zeros(shape, ref: Tensor) -> Tensor:
    t = torch.empty(shape, device=ref.device, dtype=torch.float32)
    return kern.zero_(t)


class _ZeroWs:
    """Persistent fp32 accumulators that are ZERO at rest, keyed by (device, element count): a kernel adds into one atomically
    and kern.cast_clear rounds it to the gradient's type and zeroes it again in the same pass — instead of a zero-fill launch in
    front of every use (the dK / dV accumulator of the spatial-reduction attention backward: 7 fills per step).
    A buffer that is taken stays IN USE until give_back_as: a second taker of the same size gets a buffer of its own (up to
    `MAX_LIVE` per size; beyond that the oldest is taken to be the leftover of a pass that raised between the two launches and is
    filled again).  Single-stream contract: take -> kernel -> give_back_as run on ONE stream (the buffers carry no events); with
    branch streams enabled the key holds the stream, so the branch streams of branch_stream() each own their buffers."""
    bufs: dict = {}  # key -> [[tensor, in_use], ...]
    MAX_LIVE = 4

    @staticmethod
    def _key(dev, n):
        # the stream is part of the key only while branch streams are on (two streams really do run take -> kernel -> give_back
        # concurrently then).  Otherwise ONE set per device: GraphedStep warms up on a side stream and torch.cuda.graph captures on
        # another, and a per-stream key made the capture allocate fresh buffers inside the graph's pool with their zero-fill
        # captured as a node that re-ran on every replay — the launches this class exists to avoid
        sid = torch.cuda.current_stream(dev).stream_id if (dev.type == "cuda" and _BRANCH_ON[0]) else 0
        return (dev.type, dev.index, n, sid)

    @staticmethod
    def take(shape, ref: Tensor) -> Tensor:
        n = 1
        for v in shape:
            n *= int(v)
        es = _ZeroWs.bufs.setdefault(_ZeroWs._key(ref.device, n), [])
        e = next((e for e in es if not e[1]), None)
        if e is None:
            if len(es) < _ZeroWs.MAX_LIVE:
                e = [kern.zero_(torch.empty(n, device=ref.device, dtype=torch.float32)), False]
                es.append(e)
            else:  # every buffer of this size is marked in use: leftovers of passes that raised
                e = es.pop(0)
                es.append(e)
                kern.zero_(e[0])
        e[1] = True
        return e[0].view(shape)

    @staticmethod
    def release(ws: Tensor):
        """ws is zero again (a kernel of the caller cleared it)"""
        for e in _ZeroWs.bufs.get(_ZeroWs._key(ws.device, ws.numel()), ()):
            if e[0].data_ptr() == ws.data_ptr():
                e[1] = False

    @staticmethod
    def give_back_as(ws: Tensor, like: Tensor, bias: Optional[Tensor] = None) -> Tensor:
        """-> a tensor of like's dtype (bf16) holding ws (+ bias over the last axis); ws is zero again"""
        out = torch.empty(ws.shape, device=ws.device, dtype=like.dtype)
        kern.cast_clear(ws, out, bias)
        for e in _ZeroWs.bufs.get(_ZeroWs._key(ws.device, ws.numel()), ()):
            if e[0].data_ptr() == ws.data_ptr():
                e[1] = False
        return out


def grad_buf(p: Optional[Tensor]) -> Optional[Tensor]:
    """fp32 buffer that gradient kernels ADD into for parameter `p` (None if p is frozen / absent)."""
    if p is None or not p.requires_grad:
        return None
    if p.grad is None:
        # a parameter that lives in a ParamArena gets its arena slot back (a caller ran zero_grad(set_to_none=True) or set
        # .grad = None): FusedSGD reads the arena, a free-standing buffer would leave it with a stale gradient
        home = getattr(p, "_cenet_grad_home", None)
        p.grad = home() if home is not None else _zeros(p.shape, p)
    return p.grad


def merged_param(ps, flat=False):
    """[p_0 | p_1 | ...] as ONE tensor of shape [n, *p.shape] (flat: [n * p.shape[0], *p.shape[1:]], the members stacked along
    their first axis) if the equally shaped parameters `ps` lie back to back in a
    ParamArena (its `arena_groups` layout), else None.  The result aliases the parameters' memory, its `.grad` aliases
    their gradient slots and its bf16 shadow their shadow slots, so the kernels (and ops.grad_buf / kern.wq) treat it like
    a parameter; nothing is registered anywhere and the optimizer keeps seeing the flat arena."""
    p0 = ps[0]
    slot = getattr(p0, "_cenet_arena_slot", None)
    if slot is None or any(q.shape != p0.shape or not q.requires_grad for q in ps):
        return None
    arena, o = slot
    n = p0.numel()
    for j, q in enumerate(ps):
        sj = getattr(q, "_cenet_arena_slot", None)
        if sj is None or sj[0] is not arena or sj[1] != o + j * n or q.data_ptr() != arena.params.data_ptr() + 4 * (o + j * n):
            return None
    shape = ((len(ps) * p0.shape[0],) + tuple(p0.shape[1:])) if flat else ((len(ps),) + tuple(p0.shape))
    m = arena.params[o:o + len(ps) * n].view(shape).detach()
    m.requires_grad_(True)
    m.grad = arena.grads[o:o + len(ps) * n].view(shape)
    m._cenet_grad_home = arena._home(o, len(ps) * n, shape)
    arena.enable_shadow()  # (its bf16 slots are rewritten by the fused SGD kernel: a private shadow of `m` would go stale)
    m._cenet_shadow = arena.shadow[o:o + len(ps) * n].view(shape)
    m._cenet_shadow_ver = m._version
    m._cenet_members = tuple(ps)
    return m


def refresh_member_shadows(m, like):
    """bf16 activations: a member of the merged parameter `m` that was modified through torch since its shadow slot was
    written (load_state_dict, a torch optimizer) gets the slot re-cast — the slots are shared with `m`'s shadow"""
    if like.dtype == torch.bfloat16:
        for q in m._cenet_members:
            if getattr(q, "_cenet_shadow_ver", None) != q._version:
                kern.wq(q, like)


def merged_buffer(bs):
    """the equally shaped buffers `bs` (BatchNorm running statistics / counters of sibling modules) re-homed into one
    [n, *shape] tensor; each module keeps its own tensor object, now a view of the joint one (load_state_dict copies in
    place).  Returns the joint tensor; call again if the buffers were moved since (`.to()` breaks the aliasing)."""
    b0 = bs[0]
    n = b0.numel()
    esz = b0.element_size()
    if all(q.shape == b0.shape and q.data_ptr() == b0.data_ptr() + j * n * esz for j, q in enumerate(bs)):
        base = getattr(b0, "_cenet_joint", None)
        if base is not None and base.data_ptr() == b0.data_ptr() and base.numel() == len(bs) * n:
            return base
    joint = torch.stack([q.detach() for q in bs]).contiguous()
    for j, q in enumerate(bs):
        q.data = joint[j]
    b0._cenet_joint = joint
    return joint


class _WgradCfg:
    """process-wide SWITCHES of the weight-gradient machinery (configuration, not per-pass state)"""
    overlap = False  # weight-gradient kernels on a second HIP stream (set_wgrad_overlap)
    grouping = os.environ.get("CENET_WGRAD_GROUP", "1") != "0"  # record + one grouped launch (0: the per-layer launches of round 2)
    hold_bytes = int(float(os.environ.get("CENET_WGRAD_HOLD_MB", "3072")) * (1 << 20))  # recorded operands kept alive at most
    hold = False  # measurement aid (wgrad_hold): no automatic flush, the caller flushes
    prescale = os.environ.get("CENET_LN_PRESCALE", "1") != "0"  # LayerNorm backward also writes the DropPath-scaled gradient
    # the GROUPED launches of a flush on the weight-gradient stream (round 5): ten chip-filling launches per step that only the
    # optimizer waits for, beside a backward chain of small latency-bound kernels — unlike the ~150 per-layer launches of round 2
    # (whose fork / join edges cost more under replay than the overlap bought) this is five forks per step
    flush_side = os.environ.get("CENET_WGRAD_FLUSH_SIDE", "0") != "0"


class _WgradState:
    """Per-DEVICE state of the weight-gradient machinery: the recorded (deferred) problems of the grouped launch and the
    weight-gradient stream.  One object per device index (`_wg`), created on first use: two models, or the replicas of a
    multi-device nn.DataParallel (one autograd thread per device), never share a queue, a stream or an event ring."""

    def __init__(self, device):
        self.device = device
        # --- deferred, GROUPED weight gradients (bf16 mode): LinearFn / MultiLinearFn / Conv1x1Fn / PvtMlpFn do not launch their
        # dW = dY^T X contraction; they record it here, and flush() reduces everything recorded so far with one launch per <= 56
        # problems (kern.wgrad_group, gemm_group.hip).  The queue flushes itself at the end of the backward pass (an autograd-engine
        # callback queued with the first record OF THAT PASS), and earlier wherever somebody needs the gradients: a gradient-arena
        # segment becoming final (GradReducer.segment_ready), wgrad_join() (FusedSGD.step, ParamArena.zero_grad), or when the
        # recorded operands exceed _WgradCfg.hold_bytes.  The recorded tensors are kept alive until the flush has been issued.
        self.items = []    # descriptor tuples for kern.wgrad_group
        self.keep = []     # the dY / X tensors the descriptors point into
        self.held = 0      # their bytes
        self.task = None   # autograd graph task whose end-of-backward callback is queued
        self.ln_items = []  # (partial buffer, dgamma, dbeta) of LayerNorm backward launches: folded by ONE launch at the flush
        self.rec_streams = []  # the streams the recorded operands were produced on (branch_stream(): more than one)
        # --- the weight-gradient stream
        self.stream = None
        self.events = None
        self.next_event = 0
        self.pending = False
        self.side_keep = []  # tensors the side stream still reads (see _wgrad_side): released in wgrad_join()
        # --- gradients that their producer already scaled (_prescaled_put / _prescaled_take): {data_ptr: (g, bscale, bscale * g)}.
        # Per device like everything else here: two devices' autograd threads (nn.DataParallel replicas, two models) never touch
        # the same dictionary
        self.prescaled = {}

    def flush(self):
        if not self.items and not self.ln_items:
            return
        items, self.items = self.items, []
        ln, self.ln_items = self.ln_items, []
        self.held = 0
        if self.rec_streams:
            # operands recorded from a branch stream: the flushing stream is ordered after that stream, and the allocator is told
            # that the operands are read here as well
            cur = torch.cuda.current_stream(self.device)
            other = [r for r in self.rec_streams if r != cur]
            self.rec_streams = []
            for r in other:
                cur.wait_stream(r)
            if other:
                for pair in self.keep:
                    for t in pair:
                        t.record_stream(cur)
                for part, _, _ in ln:
                    part.record_stream(cur)

        def go():
            if ln:
                kern.ln_fold_group(ln)
            if items:
                kern.wgrad_group(items, self.device)
        side = _WgradCfg.flush_side and self.device.type == "cuda" and not kern._lib.is_hostsim()
        try:
            if side:
                cur = torch.cuda.current_stream(self.device)
                if self.stream is None:
                    self.stream = torch.cuda.Stream(self.device)
                    self.events = [torch.cuda.Event() for _ in range(64)]
                ev = self.events[self.next_event & 63]
                self.next_event += 1
                ev.record(cur)
                self.stream.wait_event(ev)
                with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                    go()
                self.side_keep.extend(self.keep)  # (operands stay alive until wgrad_join: the side stream still reads them)
                self.pending = True
            elif self.device.type == "cuda" and torch.cuda.current_device() != self.device.index:
                with torch.cuda.device(self.device):
                    go()
            else:
                go()
        finally:
            self.keep = []


_WG = {}

# ---- branch streams: independent sub-graphs of the network on a second HIP stream -------------------------------------------
# The out head's 5x5 residual block reads only the input image (out.py:69): it can run on branch stream 0 beside the decoder, whose
# small-map kernels leave most of the chip idle; autograd runs each backward node on its forward's stream, so the block's backward
# (weight gradients only: the image needs no gradient) overlaps the rest of the backward pass the same way (BranchGate chooses
# where).  Under hipGraph capture the fork / join become graph edges.
# OFF by default — measured, round 5, same box, ACDC step under hipGraph replay: 18.72 / 18.78 ms without, 19.00 / 19.00 ms with
# the branch beside the decoder (backward held to x4 or not), 19.01 ms with it beside encoder stage 1.  The replayed graph then
# spreads its nodes over two hardware queues (486 / 372 kernels in the trace) and pays a cross-queue dependency at every hand-over;
# that costs more than the ~0.9 ms of residual-block work hidden.  set_branch_streams(True) / CENET_BRANCH_STREAMS=1 turn it on.
_BRANCH: dict = {}
_BRANCH_ON = [os.environ.get("CENET_BRANCH_STREAMS") is not None]


def set_branch_streams(on: bool) -> bool:
    old, _BRANCH_ON[0] = _BRANCH_ON[0], bool(on)
    return old


def branch_stream(ref: Tensor, i: int = 0):
    """branch stream i of ref's device, or None where there are no streams (CPU tensors, the host checker) or it is disabled"""
    if not _BRANCH_ON[0] or not ref.is_cuda or kern._lib.is_hostsim():
        return None
    key = (ref.device.index, i)
    st = _BRANCH.get(key)
    if st is None:
        st = _BRANCH[key] = torch.cuda.Stream(ref.device)
    return st


def branch_gate_enabled() -> bool:
    return os.environ.get("CENET_BRANCH_GATE", "1") != "0"


class _HoldFn(Function):
    @staticmethod
    def forward(ctx, t, gate):
        ctx.gate = gate
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        ev = ctx.gate.event
        if ev is not None and g.is_cuda:
            torch.cuda.current_stream(g.device).wait_event(ev)  # (this node runs on its forward's stream: the branch stream)
        return g, None


class BranchGate:
    """Holds the BACKWARD of a branch-stream sub-graph until some other point of the backward pass has been reached:
    `y = gate.hold(y)` at the end of the branch (on the branch stream), `t.register_hook(gate.release)` on the tensor whose
    gradient marks that point.  Without the release the hold does nothing."""

    def __init__(self):
        self.event = None

    def hold(self, t: Tensor) -> Tensor:
        return _HoldFn.apply(t, self)

    def release(self, grad):
        if grad.is_cuda:
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(grad.device))
        return None


def branch_streams(device) -> list:
    """the branch streams created so far on `device` (GradReducer orders the collective after them)"""
    return [s for (d, _), s in _BRANCH.items() if d == device.index]


def _wg(device) -> _WgradState:
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    st = _WG.get(key)
    if st is None:
        st = _WG[key] = _WgradState(torch.device(*key) if key[0] == "cuda" else torch.device(key[0]))
    return st


def set_wgrad_overlap(on: bool) -> bool:
    """Weight-gradient kernels can run on a second HIP stream, concurrently with the data-gradient chain of the backward
    pass: they only feed the optimizer, and most of them are short, latency-bound launches that leave the chip half idle.
    Off by default (bench.py turns it on for eager launches); `wgrad_join()` makes the current stream wait for them and must
    run before anything reads the gradients (FusedSGD.step and the gradient all-reduce call it)."""
    old = _WgradCfg.overlap
    _WgradCfg.overlap = bool(on)
    return old


@contextlib.contextmanager
def _wgrad_side(*reads, returned=None):
    """runs the body on the weight-gradient stream, ordered after everything issued so far on the current stream.
    returned: the one tensor among `reads` that the calling Function also RETURNS as a gradient (see below)"""
    if not _WgradCfg.overlap or kern._lib.is_hostsim():
        yield
        return
    dev_of = next((t.device for t in reads if isinstance(t, Tensor)), None)
    st = _wg(dev_of if dev_of is not None else torch.device("cuda", torch.cuda.current_device()))
    cur = torch.cuda.current_stream(st.device)
    if st.stream is None:
        st.stream = torch.cuda.Stream(st.device)
        st.events = [torch.cuda.Event() for _ in range(64)]  # reused round-robin: creating one per op costs more
    side = st.stream
    ev = st.events[st.next_event & 63]
    st.next_event += 1
    ev.record(cur)
    side.wait_event(ev)
    with torch.cuda.stream(side):
        yield
    for t in reads:
        if isinstance(t, Tensor):
            t.record_stream(side)  # keep the caching allocator from recycling these while the side stream reads them
    if returned is not None:
        # a gradient tensor that the Function also RETURNS (the residual gradient of Conv1x1Fn / LinearFn) would be accumulated
        # into IN PLACE on the main stream by autograd when it has a second consumer (use_count == 1 lets the engine steal the
        # buffer) while the side stream is still reading it: with an extra reference autograd accumulates out of place.  Only
        # that tensor is held (everything else is covered by record_stream), until the streams are joined (wgrad_join: the
        # optimizer step, zero_grad, the gradient all-reduce); a caller that never joins (a torch optimizer, autograd.grad
        # loops) is joined here every 256 entries, so the list cannot grow without bound.
        if len(st.side_keep) >= 256:
            cur.wait_stream(side)
            st.side_keep.clear()
        st.side_keep.append(returned)
    st.pending = True


def wgrad_stream(device=None):
    """the weight-gradient stream of `device` (default: the current one) if kernels may be pending on it, else None (for
    consumers that order another stream after it without stalling the compute stream, e.g. the gradient all-reduce)"""
    if kern._lib.is_hostsim() or not torch.cuda.is_available():
        return None
    st = _wg(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))
    return st.stream if st.pending else None


def wgrad_join():
    """every recorded weight gradient is launched and every device's compute stream waits for its weight-gradient stream"""
    for st in list(_WG.values()):
        st.flush()
        if st.pending:
            torch.cuda.current_stream(st.device).wait_stream(st.stream)
            st.pending = False
        st.side_keep.clear()


def set_wgrad_grouping(on: bool) -> bool:
    """measurement / test aid: off = every weight gradient is its own launch again (the round-2 path)"""
    wgrad_flush()
    old = _WgradCfg.grouping
    _WgradCfg.grouping = bool(on)
    return old


def wgrad_hold(on: bool):
    """measurement aid (bench.py's instrumented passes): while on, recorded weight gradients are launched only by an explicit
    wgrad_flush() / wgrad_join(), never by the end-of-backward callback"""
    _WgradCfg.hold = bool(on)


def wgrad_pending() -> int:
    """recorded, not yet launched weight-gradient problems over all devices"""
    return sum(len(st.items) + len(st.ln_items) for st in _WG.values())  # (st.keep holds the operands of st.items: not counted again)


def _wgrad_deferrable(M: int, N: int, *ts, K: int = 1, nkb: int = 1) -> bool:
    """the grouped launch takes this problem (the limits of gemm_group.hip's grp_ok: 16-bit M / N / K-batch counts, K < 2^28;
    anything else keeps its own per-layer GEMM launch instead of failing the whole end-of-backward flush)"""
    return bool(_WgradCfg.grouping and 48 <= M <= 65535 and 48 <= N <= 65535 and 1 <= nkb <= 65535 and K < (1 << 28)
                and all(t.dtype == torch.bfloat16 and t.numel() < (1 << 31) for t in ts))


_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)


def _wgrad_defer(A: Tensor, a_off: int, lda: int, skbA: int, B: Tensor, b_off: int, ldb: int, skbB: int, dW: Tensor, c_off: int,
                 db: Optional[Tensor], M: int, N: int, K: int, nkb: int, kfast: int):
    """record dW[c_off:][M, N] += sum_{kb, k} A(m, k) B(k, n), db[m] += sum A(m, k) (offsets in elements)"""
    st = _wg(A.device)
    tid = _defer_begin(st)
    st.items.append((A.data_ptr() + 2 * a_off, B.data_ptr() + 2 * b_off, dW.data_ptr() + 4 * c_off,
                     db.data_ptr() if db is not None else None, lda, ldb, skbA, skbB, M, N, K, nkb, kfast))
    st.keep.append((A, B))
    st.held += A.numel() * A.element_size() + B.numel() * B.element_size()
    _defer_end(st, tid)


def _defer_begin(st) -> int:
    """before a record: the current autograd graph task; drops the records of a pass that never reached its end"""
    tid = _graph_task_id() if _graph_task_id is not None else 0
    if st.device.type == "cuda" and _BRANCH:
        cur = torch.cuda.current_stream(st.device)
        if cur not in st.rec_streams:
            st.rec_streams.append(cur)
    if not _WgradCfg.hold and (st.items or st.ln_items) and st.task is not None and tid != st.task:
        # records of a backward pass that never reached its end (it raised: the engine runs no callbacks then).  Their
        # gradients are void; adding them into a later pass's would be wrong, and they must not block that pass's own callback.
        st.items, st.ln_items, st.keep, st.held, st.task = [], [], [], 0, None
    return tid


def _defer_end(st, tid: int):
    """after a record: make sure the end-of-backward flush of THIS pass is queued (or flush now when there is no pass)"""
    if _WgradCfg.hold:
        return
    if tid < 0:  # not inside a backward pass (a Function's backward called by hand): flush right away
        st.flush()
        return
    if st.task != tid:
        st.task = tid
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda st=st, tid=tid: _wgrad_flush_cb(st, tid))
        except RuntimeError:
            st.task = None
            st.flush()
            return
    if st.held > _WgradCfg.hold_bytes:  # bound the operands kept alive (a full launch costs nothing extra)
        st.flush()


def _ln_bwd(g, x, gamma, mean, rstd, dx, dg, db, rows, Cn, dx_add=None, up_scale=None):
    """LayerNorm backward; bf16 rows with grouping on: the affine gradients go to a partial buffer that is folded, together with
    those of every other LayerNorm of the backward segment, by one launch at the flush (no float atomics, ~1 000 single-step
    workgroups instead of ~256 x 6 - 8 dependent steps).
    up_scale: the DropPath scale of the branch that produced x (see _prescaled_put): the same kernel also leaves up_scale * dx"""
    if _WgradCfg.grouping and kern.layernorm_bwd_part_supported(g, x, Cn) and (dx_add is None or dx_add.dtype == g.dtype):
        st = _wg(g.device)
        tid = _defer_begin(st)
        dxs = None
        if up_scale is not None and _WgradCfg.prescale and rows % up_scale.numel() == 0:
            dxs = torch.empty_like(dx)
            _prescaled_put(dx, up_scale, dxs)
        part = kern.layernorm_bwd_part(g, x, gamma, mean, rstd, dx, rows, Cn, dx_add=dx_add,
                                       bscale=up_scale if dxs is not None else None, dxs=dxs)
        st.ln_items.append((part, dg, db))
        st.held += part.numel() * 4
        _defer_end(st, tid)
    else:
        kern.layernorm_bwd(g, x, gamma, mean, rstd, dx, dg, db, rows, Cn, dx_add=dx_add)


def _wgrad_flush_cb(st, tid):
    if st.task == tid:
        st.task = None
    st.flush()
    st.prescaled.clear()  # (records nobody asked for)


# ---- gradients that their producer already scaled ---------------------------------------------------------------------------
# A PVT block is x + s_b * branch(LN(x)) (pvtv2.py:141-149, s_b the DropPath scale fused into the proj / fc2 epilogue): the
# branch's backward starts from s_b * g, where g is the gradient of the block output — which the LayerNorm backward of the NEXT
# half block produces.  That kernel writes the scaled copy as a second output (one more store, no launch, no extra read) and
# leaves it here; LinearFn.backward picks it up instead of launching scale_batch.  Keyed by the storage of g; the entry holds g
# itself, so its memory cannot be handed to another tensor while the entry exists, and a pointer match means the same tensor.
# Nothing found (fp32 mode, autograd summed two gradients into a new tensor, a hook replaced it): the scale pass runs as before.
# The records live in the per-device state (_WgradState.prescaled): one dictionary per device, touched only by that device's
# autograd thread, emptied by the end-of-backward callback of the pass that filled it.
def _prescaled_put(g: Tensor, bscale: Tensor, gs: Tensor):
    d = _wg(g.device).prescaled
    if len(d) > 64:  # (records nobody asked for: a backward pass that raised before its end-of-backward callback)
        d.clear()
    d[g.data_ptr()] = (g, bscale, gs)


def _prescaled_take(g: Tensor, bscale: Tensor) -> Optional[Tensor]:
    e = _wg(g.device).prescaled.pop(g.data_ptr(), None)
    if e is None or e[0].shape != g.shape or e[0].dtype != g.dtype or e[0].stride() != g.stride():
        return None
    if e[1].data_ptr() != bscale.data_ptr() or e[1].numel() != bscale.numel():
        return None
    return e[2]


def tag_bscale(y: Tensor, bscale: Optional[Tensor]) -> Tensor:
    """y = x + bscale_b * branch: remember the scale on the tensor object so that the LayerNorm that reads y next can hand the
    branch its scaled gradient (layernorm / layernorm_res look for the tag)"""
    if bscale is not None:
        y._cenet_bscale = bscale
    return y


def wgrad_flush():
    """launch every recorded weight gradient on the current stream of its device (no-op when nothing is recorded)"""
    for st in list(_WG.values()):
        st.flush()


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return t if t is None or t.is_contiguous() else t.contiguous()


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


def _gb(p, ref):
    """gradient buffer of parameter p, or a scratch of its shape when p is frozen"""
    g = grad_buf(p)
    return g if g is not None else _zeros(p.shape, ref)


def bn_momentum(bn) -> float:
    """the running-statistics factor of a BatchNorm container.  momentum=None means a CUMULATIVE moving average in PyTorch (factor
    1 / num_batches_tracked); no kernel here implements that and the reference never builds such a layer (every BatchNorm of
    src/networks/cenet keeps the default 0.1) — refuse loudly rather than train with a silently different factor"""
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm(momentum=None) (cumulative moving average) is not supported by the HIP kernels; "
                                  "the reference network uses momentum=0.1 everywhere")
    return float(bn.momentum)


_mom = bn_momentum


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


# =====================================================================================================
# CCU and SRM gates (cfam.py:251-264, 93-101)
# =====================================================================================================
class _Batch1:
    """cfam.py:260: CCU applies its BatchNorm1d only `if B > 1`, so the reference's slice-by-slice evaluation
    (metrics_eval.py:46-49, batch 1) never runs it.  Inside `batch1_semantics()` a batch of B slices is computed as B
    independent batch-1 forwards would be (the only batch-dependent op of the eval-mode network)."""
    on = False


@contextlib.contextmanager
def batch1_semantics(on: bool = True):
    old = _Batch1.on
    _Batch1.on = bool(on)
    try:
        yield
    finally:
        _Batch1.on = old


class CCUFn(Function):
    """tap: also returns x itself; x's other consumer (the MCA shortcut, cfam.py:298-303) reads the tap and its gradient is added by
    this node's data-gradient kernel"""

    @staticmethod
    def forward(ctx, x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap=False):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        u = _empty((B, Cn, 3), x)
        amax = _empty((B, Cn), x, torch.int32)
        z = _empty((B, Cn), x)
        kern.ccu_stats_fwd(x, fc1, fc2, u, amax, z, B, Cn, HW)
        use_bn = B > 1 and not _Batch1.on
        mean = var = None
        if use_bn:
            zn = torch.empty_like(z)
            if training and kern.bn1d_supported(B):
                # [B, C] fp32 with one value per image and channel: statistics + running update + normalisation in ONE launch
                mean, var = _empty((Cn,), x), _empty((Cn,), x)
                kern.bn1d_train_fwd(z, zn, mean, var, bn_rm, bn_rv, 0.1, bn_nbt, 1e-5, bn_w, bn_b, B, Cn)
            else:
                if training:
                    mean, var = _empty((Cn,), x), _empty((Cn,), x)
                    ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
                    kern.bn_stats(z, Cn, B, Cn, 1, ws, mean, var, bn_rm, bn_rv, 0.1, bn_nbt)
                else:
                    mean, var = bn_rm, bn_rv
                kern.bn_apply(z, Cn, zn, Cn, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, Cn, 1)
        else:
            zn = z
        y = torch.empty_like(x)
        kern.gate_chan_fwd(x, zn, y, B * Cn, HW)
        ctx.save_for_backward(x, fc1, fc2, u, amax, z, zn, mean, var, bn_w, bn_b)
        ctx.refs = (fc1, fc2, bn_w, bn_b)
        ctx.cfg = (use_bn, training)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, g, g_tap=None):
        x, fc1, fc2, u, amax, z, zn, mean, var, bn_w, bn_b = ctx.saved_tensors
        use_bn, training = ctx.cfg
        if g is None:
            return (g_tap,) + (None,) * 9
        g = _c(g)
        if g_tap is not None:
            g_tap = _c(g_tap) if g_tap.dtype == g.dtype else _c(g_tap.to(g.dtype))
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        dzn = _empty((B, Cn), x)
        kern.gate_chan_bwd_reduce(x, g, zn, dzn, B * Cn, HW)
        if use_bn:
            if not training:
                raise RuntimeError("CCU backward needs training-mode BatchNorm")
            dz = torch.empty_like(dzn)
            dg, db = grad_buf(ctx.refs[2]), grad_buf(ctx.refs[3])
            if kern.bn1d_supported(B):
                kern.bn1d_bwd(dzn, z, dz, mean, var, 1e-5, bn_w, dg, db, B, Cn)  # (one launch; NULL gradients: frozen affine)
            else:
                ws = _empty((2 * Cn * 256,), x)  # CENET_BN_WS_FLOATS(C)
                if dg is None:
                    dg, db = _zeros((Cn,), x), _zeros((Cn,), x)
                kern.bn_bwd(dzn, Cn, z, Cn, dz, Cn, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, Cn, 1, ws, dg, db)
        else:
            dz = dzn
        d1, d2 = grad_buf(ctx.refs[0]), grad_buf(ctx.refs[1])
        if d1 is None:
            d1, d2 = _zeros(fc1.shape, x), _zeros(fc2.shape, x)
        dx = torch.empty_like(x)
        kern.ccu_bwd_apply(x, g, zn, dz, u, amax, fc1, fc2, d1, d2, dx, B, Cn, HW, dx_add=g_tap)
        return (dx,) + (None,) * 9


def ccu(x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap=False):
    """tap=True returns (y, x_tap): hand x_tap (not x) to x's other consumer"""
    return CCUFn.apply(x, fc1, fc2, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training, tap)


class SRMFn(Function):
    """cfam.py:93-101.  Training mode on planes of <= 4096 pixels (round 5): the conv + GELU kernel also leaves per-workgroup
    (count, mean, M2) triples, the gate kernel folds them and normalises inline, and backwards ONE kernel does BatchNorm backward,
    GELU' and the conv backward: 3 + 4 launches instead of 6 + 6 (csrc/stats.hip)."""

    @staticmethod
    def forward(ctx, x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training):
        x = _c(x)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        u = _empty((B, 3, H, Wd), x)
        amax = _empty((B, HW), x, torch.int32)
        kern.srm_stats_fwd(x, u, amax, B, Cn, HW)
        f = _empty((B, 1, H, Wd), x)
        fa = torch.empty_like(f)
        fb = torch.empty_like(f)
        y = torch.empty_like(x)
        fused = bool(training) and kern.srm_fused_supported(B, H, Wd)
        if fused:
            G = kern.srm_parts(B, H, Wd)
            part = _empty((G, 3), x)
            mean, var = _empty((1,), x), _empty((1,), x)
            kern.srm_conv_gelu_fwd(u, pwc, dwc, f, fa, part, B, H, Wd)
            kern.gate_pix_bn_fwd(x, fa, part, G, fb, y, bn_w, bn_b, 1e-5, mean, var, bn_rm, bn_rv, 0.1, bn_nbt, B, Cn, HW)
        else:
            kern.srm_conv_fwd(u, pwc, dwc, f, B, H, Wd)
            kern.act_fwd(f, fa, f.numel(), "gelu")
            if training:
                mean, var = _empty((1,), x), _empty((1,), x)
                ws = _empty((2 * 256,), x)  # CENET_BN_WS_FLOATS(1)
                kern.bn_stats(fa, HW, B, 1, HW, ws, mean, var, bn_rm, bn_rv, 0.1, bn_nbt)
            else:
                mean, var = bn_rm, bn_rv
            kern.bn_apply(fa, HW, fb, HW, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, 1, HW)
            kern.gate_pix_fwd(x, fb, y, B, Cn, HW)
        ctx.save_for_backward(x, pwc, dwc, u, amax, f, fa, fb, mean, var, bn_w, bn_b)
        ctx.refs = (pwc, dwc, bn_w, bn_b)
        ctx.training = training
        ctx.fused = fused
        return y

    @staticmethod
    def backward(ctx, g):
        x, pwc, dwc, u, amax, f, fa, fb, mean, var, bn_w, bn_b = ctx.saved_tensors
        if not ctx.training:
            raise RuntimeError("SRM backward needs training-mode BatchNorm")
        g = _c(g)
        B, Cn, H, Wd = x.shape
        HW = H * Wd
        dfb = torch.empty_like(f)
        kern.gate_pix_bwd_reduce(x, g, fb, dfb, B, Cn, HW)
        dg, db = grad_buf(ctx.refs[2]), grad_buf(ctx.refs[3])
        if dg is None:
            dg, db = _zeros((1,), x), _zeros((1,), x)
        du = torch.empty_like(u)
        dp, dd = grad_buf(ctx.refs[0]), grad_buf(ctx.refs[1])
        if dp is None:
            dp, dd = _zeros(pwc.shape, x), _zeros(dwc.shape, x)
        if ctx.fused:
            part2 = _empty((kern.srm_parts(B, H, Wd), 2), x)
            kern.srm_conv_bn_bwd(u, dfb, fa, f, mean, var, 1e-5, bn_w, pwc, dwc, part2, du, dp, dd, dg, db, B, H, Wd)
        else:
            dfa = torch.empty_like(f)
            ws = _empty((2 * 256,), x)  # CENET_BN_WS_FLOATS(1)
            kern.bn_bwd(dfb, HW, fa, HW, dfa, HW, mean, var, 1e-5, bn_w, bn_b, "none", 0.0, B, 1, HW, ws, dg, db)
            df = torch.empty_like(f)
            kern.act_bwd(f, dfa, df, f.numel(), "gelu")
            kern.srm_conv_bwd(u, df, pwc, dwc, du, dp, dd, B, H, Wd)
        dx = torch.empty_like(x)
        kern.srm_bwd_apply(x, g, fb, u, du, amax, dx, B, Cn, HW)
        return (dx,) + (None,) * 8


def srm(x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training):
    return SRMFn.apply(x, pwc, dwc, bn_w, bn_b, bn_rm, bn_rv, bn_nbt, training)


# =====================================================================================================
# DSEB combine (dseb.py:40-50,63-76,156-163)
# =====================================================================================================
class DsebCombineFn(Function):
    """z = ycoef*y + w[c]*edge(y, recon_s) + diff*y; `recons[s]` is None for scale 1.0 (e_s == 0); diff may be None."""

    @staticmethod
    def forward(ctx, y, w, diff, ycoef, n, *recons):
        y, diff = _c(y), _c(diff)
        recons = [_c(r) for r in recons]
        B, Cn = y.shape[:2]
        HW = y.numel() // (B * Cn)
        z = torch.empty_like(y)
        kern.dseb_combine_fwd(y, recons, n, w, diff, ycoef, z, B, Cn, HW)
        ctx.save_for_backward(y, w, diff, *[r for r in recons if r is not None])
        ctx.ycoef = ycoef
        ctx.mask = [r is not None for r in recons]
        ctx.refs = (w,)
        ctx.n = n
        return z

    @staticmethod
    def backward(ctx, g):
        y, w, diff = ctx.saved_tensors[:3]
        present = list(ctx.saved_tensors[3:])
        recons, it = [], iter(present)
        for m in ctx.mask:
            recons.append(next(it) if m else None)
        g = _c(g)
        B, Cn = y.shape[:2]
        HW = y.numel() // (B * Cn)
        dy = torch.empty_like(y)
        ddiff = torch.empty_like(y) if diff is not None else None
        drs = [torch.empty_like(y) if r is not None else None for r in recons]
        dw = grad_buf(ctx.refs[0])
        if dw is None:
            dw = _zeros(w.shape, y)
        kern.dseb_combine_bwd(y, recons, ctx.n, w, diff, ctx.ycoef, g, dy, drs, ddiff, dw, B, Cn, HW)
        return (dy, None, ddiff, None, None) + tuple(drs)


def dseb_combine(y, w, diff, recons: Sequence[Optional[Tensor]], ycoef: float = 2.0):
    return DsebCombineFn.apply(y, w, diff, ycoef, len(recons), *recons)


# =====================================================================================================
# loss (utils/core.py:44-80,161-188)
# =====================================================================================================
class DiceCELossFn(Function):
    """w_dice * Dice + w_ce * CE + w_bd * BoundaryDoU in one pass over the logits each way (core.py:44-131,161-188)."""

    @staticmethod
    def forward(ctx, logits, labels, w_dice, w_ce, w_bd=0.0):
        logits, labels = _c(logits), _c(labels)
        B, K = logits.shape[:2]
        H, W = (logits.shape[2], logits.shape[3]) if logits.dim() == 4 else (1, logits.numel() // (B * K))
        acc = _empty((16384,), logits)  # CENET_LOSS_ACC_FLOATS (include/cenet_hip.h): replicated partial sums, one value per line
        loss = _empty((1,), logits)
        kern.seg_loss_fwd(logits, labels, acc, loss, B, K, H, W, w_dice, w_ce, w_bd)
        ctx.save_for_backward(logits, labels, acc)
        ctx.cfg = (w_dice, w_ce, w_bd, H, W)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, labels, acc = ctx.saved_tensors
        w_dice, w_ce, w_bd, H, W = ctx.cfg
        B, K = logits.shape[:2]
        g = _c(g).reshape(1)
        d = torch.empty_like(logits)
        kern.seg_loss_bwd(logits, labels, acc, g, d, B, K, H, W, w_dice, w_ce, w_bd)
        return d, None, None, None, None


def dice_ce_loss(logits, labels, w_dice=0.5, w_ce=0.5, w_boundary=0.0):
    return DiceCELossFn.apply(logits, labels, w_dice, w_ce, w_boundary)
