#!/usr/bin/env python
"""What a NEW batch costs before the layers run (GPU only): both edge groupings (tfc_csr_build), the M pre-pass and
the class table, at the batch of a bench workload.   python tools/bench_topology.py [c3|c4|c5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matten_amd import ops, keys as K, _lib
from matten_amd.data.crystal import synthetic_batch
from bench import WORKLOADS, R_CUT

wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
dev = torch.device("cuda")
batch, _ = synthetic_batch(wl["n"], wl["a"], wl["graphs"], seed0=1000, r_cut=R_CUT)
ei = batch[K.EDGE_INDEX].to(dev)
N, E = batch[K.POSITIONS].shape[0], ei.shape[1]
L = _lib.lib()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3, (time.perf_counter() - t0) / n * 1e6


ops.VALIDATE_EDGES = "off"
gpu, wall = timed(lambda: ops.EdgeOrdering(ei, N))
print(f"N {N} E {E}: EdgeOrdering (two groupings) {gpu:8.1f} us GPU, {wall:8.1f} us wall")
L.tfc_profile_reset()
L.tfc_profile_enable(1)
for _ in range(5):
    ops.EdgeOrdering(ei, N)
torch.cuda.synchronize()
L.tfc_profile_enable(0)
for k, (c, tot) in sorted(_lib.profile_report().items()):
    print(f"   {k:32s} {c / 5:5.1f} x {tot / c * 1e3:8.1f} us")
cls = torch.randint(0, 4, (N,), device=dev)
cls.tfc_num_classes = 4
a = torch.randn(4, wl["emb"], device=dev)[cls]


def table():
    ops._ATTR_CLASSES.clear()
    ops._ATTR_FACTORS.clear()
    ops.attr_factors(a, cls)


gpu, wall = timed(table)
print(f"class table + distinct rows: {gpu:8.1f} us GPU, {wall:8.1f} us wall")
