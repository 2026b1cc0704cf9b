"""cenet_amd.ops.infra — shared plumbing of the operator modules: tensor allocation helpers, zero-at-rest workspaces, parameter / gradient views into the
ParamArena (merged_param), the recorded weight-gradient queue and its per-device state, branch streams, pre-scaled gradient hand-over.

(ops package overview) Autograd operators of the CENet hot path, forward AND backward on hand-written HIP kernels.

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

from .. import kern

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


# ---- small helpers shared by several operator modules ---------------------------------------------------------------------------
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


__all__ = [n for n in dir() if not n.startswith("__")]
