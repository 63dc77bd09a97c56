#!/usr/bin/env python
"""Micro-benchmark of the radial network alone (GPU only): forward and backward launches at the layer
shapes of configs 3 and 4, one shape per line, with the rates against the fp32 matrix peak and HBM.
    python tools/bench_mlp.py [c3|c4]      (TFC_MLP_X3=0 etc. select the kernel variants)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matten_amd import ops, _lib

dev = torch.device("cuda")
SHAPES = {"c3": (59346, [192, 960, 1728]), "c4": (200196, [512, 3584, 8192])}


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def one(E, W):
    torch.manual_seed(0)
    x = torch.randn(E, 8, device=dev)
    ws = [torch.randn(8, 64, device=dev, requires_grad=True), torch.randn(64, 64, device=dev, requires_grad=True),
          torch.randn(64, W, device=dev, requires_grad=True)]
    cst = 1.8782
    flops = 2.0 * E * (8 * 64 + 64 * 64 + 64 * W)
    byts = 4.0 * E * W
    us = timeit(lambda: ops._radial_mlp_run(x, [w.detach() for w in ws], cst))
    out = ops.radial_mlp(x, ws, cst)
    g = torch.randn_like(out)
    L = _lib.lib()
    L.tfc_profile_reset()
    L.tfc_profile_enable(1)
    for _ in range(5):
        torch.autograd.grad(out, ws, g, retain_graph=True)
    torch.cuda.synchronize()
    L.tfc_profile_enable(0)
    prof = _lib.profile_report()
    L.tfc_profile_reset()
    parts = "  ".join(f"{k} {tot / c * 1e3:8.1f} us" for k, (c, tot) in sorted(prof.items()))
    tot_b = sum(tot / c * 1e3 * (c / 5) for k, (c, tot) in prof.items())
    print(f"E {E} W {W}: fwd {us:8.1f} us {flops / us / 1e6:6.1f} TF {byts / us / 1e6:5.2f} TB/s | bwd {tot_b:8.1f} us "
          f"{2 * flops / tot_b / 1e6:6.1f} TF | {parts}")


if __name__ == "__main__":
    for cfg in sys.argv[1:] or ["c3", "c4"]:
        E, Ws = SHAPES[cfg]
        for W in Ws:
            one(E, W)
