"""Data parallelism over whole graphs: one process per GPU, one flat fp32 gradient bucket,
one all-reduce per step (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo"
on CPU for tests).

The reference has no distributed code of its own: multi-GPU is whatever Lightning's
Trainer(gpus=..., num_nodes=...) does (scripts/configs/minimal.yaml:62-64), i.e. DDP's
bucketed gradient all-reduce.  The forward/backward of the hot path needs no communication
(graphs are independent, no edge sharding), so the only exchange is this one sum.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch


def shard_graphs(n_graphs: int, rank: int, world: int) -> List[int]:
    """Graphs of a global batch owned by `rank`: {g : g mod world == rank}."""
    return [g for g in range(n_graphs) if g % world == rank]


class FlatGradBucket:
    """All parameter gradients as views into one contiguous buffer.

    `param.grad` of every parameter aliases a slice of `flat`, so autograd accumulates straight
    into the bucket and `all_reduce_mean()` is a single collective on a single tensor with no
    gather/scatter copies."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("all parameters must share device and dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off : off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(world)
