#!/usr/bin/env python
"""How much of a bench step is host time: enqueue time of K steps (no sync) vs. wall time with sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import matten_amd.nn as H
from matten_amd import keys as K
from matten_amd.harness import ConvStack
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
batch = bench.build_batch(wl, 0, dev)
torch.manual_seed(35)
model = ConvStack(H, wl["emb"], wl["lmax"], wl["irreps"], wl["layers"], 64, 2, 5.0, 28.0).to(dev)
pos0 = batch[K.POSITIONS]
def step():
    b = dict(batch); pos = pos0.detach().requires_grad_(True); b[K.POSITIONS] = pos
    _, e, _ = model(b, wl["graphs"]); e.sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
one = []
for _ in range(10):
    torch.cuda.synchronize(); ta = time.perf_counter(); step(); tb = time.perf_counter(); one.append(tb - ta)
torch.cuda.synchronize()
print(f"host time of one step issued into an empty queue: median {1e3*sorted(one)[5]:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
