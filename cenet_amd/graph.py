"""HIP-graph capture of a whole training step.

The reference has a tracing compiler option (`torch.compile(fullgraph=True)`, src/main_acdc.py:188-191); the MI355X-native
equivalent here is explicit: every kernel of the step (one memset, ~1.8 k hand-written launches, the RCCL all-reduces and
the fused SGD update) is recorded ONCE into a hipGraph on a side stream and replayed per iteration, so the host issues a
single call per step and the launch gaps disappear.  Requirements met by the rest of the package: no host synchronisation
and no host->device copies inside a step (the loss never leaves the device, `FusedSGD.prepare()` uploads the
hyper-parameters before the replay), all temporaries come from the torch caching allocator (graph-private pool),
parameter/gradient storage is static (ParamArena).
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    """Captures `step_fn()` (which must run zero_grad -> forward -> loss -> backward -> [reduce] -> optimizer.step
    (sync_hyper=False) on STATIC input tensors and return the loss tensor) and replays it."""

    def __init__(self, step_fn: Callable[[], torch.Tensor], optimizer=None, warmup: int = 3):
        self.optimizer = optimizer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # allocator / lazy-init warm-up on the capture stream
                if optimizer is not None:
                    optimizer.prepare()
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if optimizer is not None:
            optimizer.prepare()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()
        self.steps_in_capture = warmup + 1

    def __call__(self) -> torch.Tensor:
        if self.optimizer is not None:
            self.optimizer.prepare()
        self.graph.replay()
        if self.optimizer is not None:
            self.optimizer._steps += 1
        return self.loss


class GraphedSplitStep:
    """Data-parallel form: the step is captured as TWO hipGraphs with the gradient all-reduce between them, issued eagerly —

        graph A: zero_grad -> forward -> loss -> backward        (all weight gradients joined at its end)
        eager  : between()   (GradReducer.finish(): one RCCL all-reduce per arena segment, a handful of host calls)
        graph B: optimizer.step(sync_hyper=False)

    so a rank's host issues ~8 calls per step instead of ~1 300 and the step no longer depends on how fast (or how shared) the
    host is, at the price of not overlapping the 133 MB all-reduce with the tail of the backward pass (~1.5 ms over xGMI, SURVEY
    §5).  Collectives stay outside the captured regions: capturing them aborted inside RCCL on this stack."""

    def __init__(self, fwd_bwd_fn: Callable[[], torch.Tensor], optimizer, between: Callable[[], None], warmup: int = 3):
        from . import ops
        self.optimizer, self.between = optimizer, between

        def a_fn():
            loss = fwd_bwd_fn()
            ops.wgrad_join()  # the side-stream branch must re-join inside the capture
            return loss
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.prepare()
                a_fn()
                between()
                optimizer.step(sync_hyper=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.prepare()
        # (the process group's watchdog thread polls the events of collectives still in flight; an event query from another
        # thread while a capture is open is an error in the default "global" capture mode: drain first, capture thread-locally)
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self.loss = a_fn()
        between()
        torch.cuda.synchronize()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
            optimizer.step(sync_hyper=False)
        optimizer._steps -= 1  # the captured call counted itself; replays count below
        torch.cuda.synchronize()

    def __call__(self) -> torch.Tensor:
        self.optimizer.prepare()
        self.graph_a.replay()
        self.between()
        self.graph_b.replay()
        self.optimizer._steps += 1
        return self.loss


def enable_segment_cuts(net):
    """CENet / DataParallel(CENet): from the next training forward on, the four encoder stage outputs are cut out of the autograd
    graph (pvtv2.forward_features) and listed in the returned list as (stage output, leaf) pairs, stage 1 first."""
    net = getattr(net, "module", net)
    net.backbone.segment_cuts = []
    return net.backbone.segment_cuts


def disable_segment_cuts(net):
    getattr(net, "module", net).backbone.segment_cuts = None


def backward_pieces(loss: torch.Tensor, cuts):
    """The backward pass of a forward made with segment cuts, as five callables in gradient-arena order
    (cenet_amd.optim.cenet_segments): piece 0 differentiates the loss down to the cut leaves (head + decoder parameters final),
    piece k = 1..4 differentiates encoder stage 5 - k from the gradient its leaf has collected (that stage's parameters final;
    the leaf below it receives its last contribution).  Each piece is its own autograd-engine run, so the recorded (grouped)
    weight gradients of a segment are launched when its piece ends (ops._WgradState)."""
    from . import ops

    def head():
        loss.backward()
        ops.wgrad_join()  # a piece ends with all of ITS weight gradients issued on the compute stream (and, inside a capture,
                          # with the weight-gradient stream joined: a graph must not end with unjoined work)
    pieces = [head]
    for t, leaf in reversed(cuts):
        def stage(t=t, leaf=leaf):
            torch.autograd.backward([t], [leaf.grad])
            ops.wgrad_join()
        pieces.append(stage)
    return pieces


class SegmentedStep:
    """Data-parallel step that keeps the backward / all-reduce overlap under hipGraph replay (bench.py, N > 1).

        graph 0 : zero_grad -> forward -> loss -> backward piece 0 (head + decoder)
        eager   : on_segment(0)      GradReducer.segment_ready(0): the segment's all-reduce starts on the communication stream
        graph k : backward piece k   (encoder stage 5 - k), k = 1..4, each followed by on_segment(k)
        eager   : finish()           GradReducer.finish(): the compute stream waits for the five collectives
        graph 5 : optimizer.step(sync_hyper=False)

    ~15 host calls per step instead of ~1 300 eager launches, and — unlike GraphedSplitStep, which reduces everything after one
    big backward graph — segment k's collective runs beside graphs k+1..4.  Collectives stay outside the captures (capturing
    them aborted inside RCCL on this stack).  `graphs=False` runs the same pieces eagerly (CPU tests, gloo)."""

    def __init__(self, net, fwd_loss_fn: Callable[[], torch.Tensor], optimizer, on_segment: Callable[[int], None],
                 finish: Callable[[], None], graphs: bool = True, warmup: int = 2):
        self.net, self.fwd_loss_fn, self.optimizer = net, fwd_loss_fn, optimizer
        self.on_segment, self.finish = on_segment, finish
        self.graphs = None
        # the stage-output hooks of an attached GradReducer would fire inside piece 0 on decoder-only partial gradients and start
        # the stage segments' all-reduces before their backward has run: the segmented step drives the reducer itself
        core = getattr(net, "module", net)
        if getattr(core, "_grad_sync", None) is not None:
            raise RuntimeError("SegmentedStep: detach the GradReducer's backward hooks (net._grad_sync) first; pass "
                               "reducer.segment_ready / reducer.finish as on_segment / finish instead")
        if not graphs:
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                optimizer.prepare()
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        optimizer.prepare()
        self.cuts = enable_segment_cuts(net)
        try:  # whatever a capture raises, the model must not stay cut (its encoder would silently stop receiving gradients)
            self._capture(net, fwd_loss_fn, optimizer, on_segment, finish)
        except BaseException:
            self.graphs = None
            raise
        finally:
            disable_segment_cuts(net)  # replays run no Python forward; eager forwards of the same model are whole again

    def _capture(self, net, fwd_loss_fn, optimizer, on_segment, finish):
        # capture: thread-local error mode after a full drain (the process group's watchdog polls events of in-flight
        # collectives from another thread, see GraphedSplitStep)
        self.graphs = []
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, capture_error_mode="thread_local"):
            optimizer.zero_grad()
            self.loss = fwd_loss_fn()
            if len(self.cuts) != 4:
                raise RuntimeError(f"SegmentedStep: expected the four encoder stage outputs to be cut, got {len(self.cuts)} "
                                   "(the forward must run CENet.backbone.forward_features in training mode with grad enabled)")
            pieces = backward_pieces(self.loss, self.cuts)
            pieces[0]()
        self.graphs.append(g0)
        on_segment(0)
        for k in range(1, 5):
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=g0.pool(), capture_error_mode="thread_local"):
                pieces[k]()
            self.graphs.append(g)
            on_segment(k)
        finish()
        torch.cuda.synchronize()
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, pool=g0.pool(), capture_error_mode="thread_local"):
            optimizer.step(sync_hyper=False)
        optimizer._steps -= 1
        self.graphs.append(gs)
        torch.cuda.synchronize()

    def _eager(self):
        cuts = enable_segment_cuts(self.net)
        try:
            self.optimizer.zero_grad()
            loss = self.fwd_loss_fn()
            for k, piece in enumerate(backward_pieces(loss, cuts)):
                piece()
                self.on_segment(k)
        finally:
            disable_segment_cuts(self.net)
        self.finish()
        self.optimizer.step(sync_hyper=False)
        return loss

    def __call__(self) -> torch.Tensor:
        self.optimizer.prepare()
        if self.graphs is None:
            return self._eager()
        for k in range(5):
            self.graphs[k].replay()
            self.on_segment(k)
        self.finish()
        self.graphs[5].replay()
        self.optimizer._steps += 1
        return self.loss
