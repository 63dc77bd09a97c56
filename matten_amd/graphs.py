"""HIP-graph replay of a whole step (forward, backward, gradient accumulation) for callers whose batch
shape does not change between steps: molecular dynamics on one cell, repeated evaluation of one batch,
training with padded / bucketed batches.

A step of the hot path is ~170 short launches on two streams; issued one by one from Python the host needs
longer to enqueue them than the GPU needs to run the forward half.  Captured once (torch.cuda.graph: the
launches of libtfc go to torch's current stream, so they are captured like torch's own kernels, the side
stream's fork and join included) the step replays with a single host call.  The reference has no counterpart
(its step is Lightning's eager loop, eigenn/model/model.py:239-327).

Rules for the captured function: inputs and outputs live in tensors that keep their addresses (copy new
values into them between replays), no host synchronisation inside (`.item()`, data-dependent shapes), and
every cache the layers keep must be warm - GraphedStep runs the function a few times eagerly first.
"""
from __future__ import annotations

from typing import Callable

import torch


class GraphedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 3):
        from . import ops
        from .nn import point_conv

        self.fn = fn
        ops.clear_caches()  # (also synchronises the device)
        point_conv.reset_prefetch_state()
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        # (parameters that accumulated gradients in earlier eager steps keep their AccumulateGrad nodes on the stream
        # of those steps; torch warns about the mismatch, which costs one event wait per parameter and nothing else)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        with torch.cuda.stream(s):  # torch asks for the warm-up on a non-default stream
            for _ in range(warmup):
                fn()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
