"""HIP-graph replay of a whole step (forward, backward, gradient accumulation) for callers whose batch
SHAPE does not change between steps: molecular dynamics on one cell, repeated evaluation of one batch,
training with padded / bucketed batches.

A step of the hot path is ~170 short launches on two streams; issued one by one from Python the host needs
longer to enqueue them than the GPU needs to run the forward half.  Captured once (torch.cuda.graph: the
launches of libtfc go to torch's current stream, so they are captured like torch's own kernels, the side
stream's fork and join included) the step replays with a single host call.  The reference has no counterpart
(its step is Lightning's eager loop, eigenn/model/model.py:239-327).

Rules for the captured function: inputs and outputs live in tensors that keep their addresses (copy new
values into them between replays), no host synchronisation inside (`.item()`, data-dependent shapes).

What may change between replays depends on `rebuild_topology`:

* False (default): the edge orderings, the M-matrix layout and the class structure of the node attributes are
  built during the eager warm-up and are NOT part of the graph.  Only positions, cell, shifts' values and
  parameters may be updated in place; the tensors named in `static_topology` (edge_index, batch, species /
  node-attribute classes ...) must not be written to - GraphedStep remembers their versions and raises on the
  next call if one has changed (a replay would pair the new edge_index with the old orderings, silently).
* True: the caches are dropped right before the capture, so the topology work (tfc_csr_build for both
  groupings, the class table) is captured with the step and every replay redoes it from the CURRENT contents
  of the static tensors: a new batch of the same shape (same N, same padded E, same number of graphs) is
  taken by copy-in.  Requirements: the node-attribute classes come with their count (SpeciesEmbedding sets
  it; ops._class_structure_static) so that no step of it synchronises, and padded edges are null edges
  (longer than the cutoff: their radial embedding, weights, messages and gradients are exactly zero).
  Edge-index validation does not run inside a replay.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch


class GraphedStep:
    def __init__(self, fn: Callable[[], object], warmup: int = 3, static_topology: Optional[Sequence[torch.Tensor]] = None,
                 rebuild_topology: bool = False):
        from . import ops
        from .nn import point_conv

        self.fn = fn
        self.rebuild_topology = bool(rebuild_topology)
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
        try:
            with torch.cuda.stream(s):  # torch asks for the warm-up on a non-default stream
                for _ in range(warmup):
                    fn()
            cur.wait_stream(s)
            torch.cuda.synchronize()
            if self.rebuild_topology:
                # every topology cache misses inside the capture: its kernels become part of the graph
                ops.clear_caches()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = fn()
            # the graph reads cached tensors (side-stream workspaces made in the warm-up, orderings, class tables)
            # at their captured addresses: they live as long as this object does, whatever clears the caches later
            self._keepalive = ops.snapshot_caches()
        finally:
            if quiet is not None:
                quiet(True)  # the diagnostic is on by default: give it back to unrelated code
        self._static = [(t, t._version) for t in (static_topology or ())]

    def __call__(self):
        if not self.rebuild_topology:
            for t, v in self._static:
                if t._version != v:
                    raise RuntimeError(
                        "GraphedStep: a topology tensor (edge_index / batch / species ...) was modified after the capture; the "
                        "graph would replay with the orderings of the old one.  Capture with rebuild_topology=True to take new "
                        "batches by copy-in, or build a new GraphedStep."
                    )
        self.graph.replay()
        return self.out


class PackedBatch:
    """The tensors of a batch as views into ONE flat device buffer, so that a new batch of the same layout is taken by
    the static buffers of a captured step with a single device copy instead of one per tensor (positions, cell,
    edge_index, shifts, batch, species ...: ~9 launches, which at the size of config 3 cost as much host time as a third
    of the replayed step).

        static = PackedBatch(first_batch)           # the buffers the captured step reads: static.tensors
        nxt = PackedBatch(next_batch, like=static)  # the loader's side; same keys, shapes and dtypes
        static.load(nxt); replay()

    Every view starts on a 256-byte boundary (the vector loads of the kernels ask for 16)."""

    ALIGN = 256

    def __init__(self, batch: dict, like: "PackedBatch | None" = None):
        keys = sorted(batch)
        layout, off = [], 0
        for k in keys:
            v = batch[k]
            nbytes = v.numel() * v.element_size()
            layout.append((k, tuple(v.shape), v.dtype, off, nbytes))
            off += (nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if like is not None and layout != like.layout:
            raise ValueError("PackedBatch: keys, shapes or dtypes differ from the batch the static buffers were laid out for")
        self.layout = layout
        dev = batch[keys[0]].device
        self.flat = torch.zeros(max(off, 1), dtype=torch.uint8, device=dev)
        self.tensors = {}
        for k, shape, dtype, o, nbytes in layout:
            view = self.flat[o : o + nbytes].view(dtype).view(shape)
            view.copy_(batch[k].detach())
            self.tensors[k] = view

    def load(self, other: "PackedBatch"):
        """Overwrite this batch with `other` (one copy on the current stream)."""
        if other.layout != self.layout:
            raise ValueError("PackedBatch.load: layouts differ")
        with torch.no_grad():
            self.flat.copy_(other.flat)
