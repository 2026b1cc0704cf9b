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
