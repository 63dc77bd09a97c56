"""Rehearsal of the gradient exchange with N ranks sharing ONE card (gloo backend, CUDA tensors):
which torch.distributed call patterns complete.  Not part of the product or of the tests.

Finding (r01, MI355X box, torch 2.10+rocm7.0): with gloo, an all_reduce on a CUDA tensor issued
while the compute stream still has work queued never completes (both ranks block in Work.wait();
this script reproduces it with plain torch matmuls, no libtfc involved); with the device idle at the
call it completes.  So N ranks on one card cannot rehearse the overlapped gradient exchange of
matten_amd/dp.py - bench.py therefore keeps `--overlap 0` (one all-reduce after backward) as its
default until the overlapped path has been timed on a multi-GPU node over RCCL."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)
dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda", 0)


def say(msg):
    print(f"[{rank}] {time.time():.1f} {msg}", file=sys.stderr, flush=True)


t = torch.ones(1 << 20, device=dev) * (rank + 1)
dist.all_reduce(t)
torch.cuda.synchronize()
say(f"sync all_reduce ok {t[0].item()}")
w = dist.all_reduce(t, async_op=True)
w.wait()
torch.cuda.synchronize()
say(f"one async ok {t[0].item()}")
flat = torch.ones(6 << 20, device=dev)
ws = [dist.all_reduce(flat[i << 20 : (i + 2) << 20], async_op=True) for i in (0, 2, 4)]
for x in ws:
    x.wait()
torch.cuda.synchronize()
say(f"three async slices ok {flat[0].item()} {flat[-1].item()}")
dist.barrier()
torch.cuda.synchronize()
say("barrier ok")
for it in range(3):
    ws = [dist.all_reduce(flat[i << 20 : (i + 2) << 20], async_op=True) for i in (0, 2, 4)]
    for x in ws:
        x.wait()
    torch.cuda.synchronize()
    say(f"after barrier: round {it} ok")
dist.barrier()
say("second barrier ok")
# all-reduce issued while the compute stream still has work queued by this thread
a = torch.randn(8192, 8192, device=dev)
for it in range(3):
    b = a
    for _ in range(20):
        b = (b @ a) * 1e-2
    flat.copy_(b.reshape(-1)[: flat.numel()])
    ws = [dist.all_reduce(flat[i << 20 : (i + 2) << 20], async_op=True) for i in (0, 2, 4)]
    for x in ws:
        x.wait()
    torch.cuda.synchronize()
    say(f"pending work, main thread: round {it} ok")
# ... and queued by the autograd thread
w = torch.randn(4096, 4096, device=dev, requires_grad=True)
w.grad = flat[: 4096 * 4096].view(4096, 4096)
for it in range(3):
    y = w
    for _ in range(10):
        y = (y @ w) * 1e-2
    y.sum().backward()
    ws = [dist.all_reduce(flat[i << 20 : (i + 2) << 20], async_op=True) for i in (0, 2, 4)]
    for x in ws:
        x.wait()
    torch.cuda.synchronize()
    say(f"pending work, autograd thread: round {it} ok")
dist.destroy_process_group()
