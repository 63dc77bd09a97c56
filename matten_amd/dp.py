"""Data parallelism over whole graphs: one process per GPU, one flat fp32 gradient bucket,
one all-reduce per step (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo"
on CPU for tests).

The reference has no distributed code of its own: multi-GPU is whatever Lightning's
Trainer(gpus=..., num_nodes=...) does (scripts/configs/minimal.yaml:62-64), i.e. DDP's
bucketed gradient all-reduce.  The forward/backward of the hot path needs no communication
(graphs are independent, no edge sharding), so the only exchange is this one sum.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import os
import sys

import torch

_DEBUG = bool(os.environ.get("TFC_DP_DEBUG"))


def shard_graphs(n_graphs: int, rank: int, world: int) -> List[int]:
    """Graphs of a global batch owned by `rank`: {g : g mod world == rank}."""
    return [g for g in range(n_graphs) if g % world == rank]


class FlatGradBucket:
    """All parameter gradients as views into one contiguous buffer, reduced in a few large slices
    that overlap with the rest of the backward pass.

    `param.grad` of every parameter aliases a slice of `flat`, so autograd accumulates straight
    into the buffer and a collective runs on a slice of it with no gather/scatter copies.  The
    buffer is laid out in REVERSE registration order (the order in which backward finishes the
    gradients: last layer first) and cut into slices of at least `bucket_bytes`; a slice's
    all-reduce is launched (async, on the backend's own stream) from a post-accumulate hook as
    soon as its last gradient has been written, in slice order on every rank, while the earlier
    layers are still in backward (`overlap=True`; opt-in until it has been timed over RCCL).  `all_reduce_mean()` launches whatever is left (slices whose
    parameters got no gradient), waits and divides.  With `overlap=False` there are no hooks and
    the buffer is one slice: a single all-reduce after backward.

    xGMI is point to point and a ring all-reduce is per-link bound, so slices are
    kept large: the default cap of 8 MB gives 3 slices at config 3 (22 MB of gradients), of which
    only the last, smallest one (the first layer's) is exposed after backward ends."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 8 << 20, overlap: bool = False, group=None,
                 force: bool = False, alias: "bool | None" = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share device and dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.group = group
        self.overlap = overlap
        # alias=True: every p.grad is a view into `flat` for the whole run (needed to reduce slices while backward is
        # still running).  alias=False (default without overlap): gradients are left to autograd - zero() drops them,
        # so backward hands each parameter its gradient tensor as is instead of launching one `grad += new` kernel
        # per parameter into a zeroed buffer - and all_reduce_mean() packs them into `flat` with one concatenation
        # right before the collective, after which the parameters' .grad are views of the reduced buffer.
        self.alias = bool(overlap) if alias is None else bool(alias)
        if overlap and not self.alias:
            raise ValueError("overlap=True needs alias=True")
        self.force = force  # run the collectives even in a group of one rank (exercises the RCCL path on one GPU)
        self._views = {}  # id(param) -> (offset, numel): where its gradient lives in `flat`
        self.bounds: List[Tuple[int, int]] = []  # [begin, end) of each slice in `flat`
        self._slice_of = {}
        self._n_in_slice: List[int] = []
        off = begin = count = 0
        item = self.flat.element_size()
        for p in reversed(self.params):
            if self.alias:
                p.grad = self.flat[off : off + p.numel()].view_as(p)
            self._views[id(p)] = (off, p.numel())
            self._slice_of[id(p)] = len(self.bounds)
            off += p.numel()
            count += 1
            if overlap and (off - begin) * item >= bucket_bytes:  # (no overlap: one collective on the whole buffer)
                self.bounds.append((begin, off))
                self._n_in_slice.append(count)
                begin, count = off, 0
        if count:
            self.bounds.append((begin, off))
            self._n_in_slice.append(count)
        self._pending = list(self._n_in_slice)
        self._works: List[object] = [None] * len(self.bounds)
        self._next = 0
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if overlap else []

    # ---- internals
    def _active(self) -> bool:
        import torch.distributed as dist

        return dist.is_available() and dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def _aliased(self, p) -> bool:
        off, n = self._views[id(p)]
        return p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + off * self.flat.element_size() and p.grad.numel() == n

    def _launch(self, i: int):
        import torch.distributed as dist

        b, e = self.bounds[i]
        self._works[i] = dist.all_reduce(self.flat[b:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        i = self._slice_of[id(p)]
        self._pending[i] -= 1
        if not self._active():
            return
        # strictly in slice order, so that every rank issues the same sequence of collectives
        while self._next < len(self.bounds) and self._pending[self._next] == 0 and self._works[self._next] is None:
            self._launch(self._next)
            self._next += 1

    # ---- per step
    def zero(self):
        """Start of a step: clears the gradients and re-arms the slices.  Gradients that no longer alias the
        buffer (optimizer.zero_grad() sets them to None by default) are pointed back into it."""
        if not self.alias:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()
            for p in self.params:
                if not self._aliased(p):
                    off, n = self._views[id(p)]
                    p.grad = self.flat[off : off + n].view_as(p)
        self._pending = list(self._n_in_slice)
        self._works = [None] * len(self.bounds)
        self._next = 0

    def all_reduce_mean(self, group=None, grads: "Sequence[torch.Tensor] | None" = None):
        """End of backward: every gradient becomes the mean over the ranks.  `grads` (alias=False only): the
        gradient tensors to reduce, in the order of the parameters, when they are not the parameters' current
        `.grad` - a step replayed from a HIP graph writes into the tensors that were `.grad` when it was captured,
        whatever `.grad` points to now."""
        if group is not None:
            self.group = group
        if not self._active():
            return
        import torch.distributed as dist

        if not self.alias:
            # pack (one kernel), reduce, and leave every .grad a view of the reduced buffer
            order = list(reversed(self.params))
            if grads is not None:
                if len(grads) != len(self.params):
                    raise ValueError("all_reduce_mean: one gradient per parameter expected")
                src = {id(p): g for p, g in zip(self.params, grads)}
            else:
                src = {id(p): p.grad for p in self.params}
            parts = []
            for p in order:
                g = src[id(p)]
                if g is not None and g.data_ptr() == self.flat.data_ptr() + self._views[id(p)][0] * self.flat.element_size():
                    raise RuntimeError("FlatGradBucket: a gradient already lives in the flat buffer (all_reduce_mean called "
                                       "twice without a backward in between, or a replayed step without `grads=`)")
                parts.append((g if g is not None else torch.zeros_like(p)).reshape(-1))
            torch.cat(parts, out=self.flat)
            work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            work.wait()
            self.flat.div_(dist.get_world_size(self.group))
            for p in order:
                off, n = self._views[id(p)]
                p.grad = self.flat[off : off + n].view_as(p)
            return
        for p in self.params:
            if not self._aliased(p):
                # e.g. optimizer.zero_grad(set_to_none=True) after zero(): backward then wrote to a fresh tensor
                # and the buffer holds stale zeros; reducing it would silently desynchronise the ranks
                raise RuntimeError("FlatGradBucket: a parameter's .grad no longer aliases the flat buffer; call bucket.zero() "
                                   "at the start of every step instead of optimizer.zero_grad()")
        for i in range(len(self.bounds)):
            if self._works[i] is None:
                self._launch(i)
        for i, w in enumerate(self._works):
            if _DEBUG:
                print(f"[dp] wait slice {i} {self.bounds[i]}", file=sys.stderr, flush=True)
            w.wait()
        self._next = len(self.bounds)
        self.flat.div_(dist.get_world_size(self.group))
