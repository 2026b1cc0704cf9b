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
