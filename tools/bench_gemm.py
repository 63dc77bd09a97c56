#!/usr/bin/env python
"""Micro-benchmark of tfc_gemm on the shapes the layer produces (GPU only)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matten_amd import ops

dev = torch.device("cuda")

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

def plain(m, n, k, layout="nn"):
    A = torch.randn(m, k, device=dev); B = torch.randn(k, n, device=dev); c = torch.empty(m, n, device=dev)
    if layout == "nn":
        a, b = A, B; a_r, a_k, b_k, b_j = ops._lin(k), ops._lin(1), ops._lin(n), ops._lin(1)
    elif layout == "nt":
        a, b = A, B.t().contiguous(); a_r, a_k, b_k, b_j = ops._lin(k), ops._lin(1), ops._lin(1), ops._lin(k)
    else:
        a, b = A.t().contiguous(), B; a_r, a_k, b_k, b_j = ops._lin(1), ops._lin(m), ops._lin(n), ops._lin(1)
    f = lambda: ops._gemm(m, n, k, ops._p(a), ops._p(b), ops._p(c), a_r, a_k, b_k, b_j, ops._lin(n), ops._lin(1))
    us = timeit(f)
    print(f"plain {layout} {m}x{n}x{k}: {us:9.1f} us  {2*m*n*k/us/1e6:7.1f} TF/s  C-write {m*n*4/us/1e6:6.2f} TB/s")

def fctp(n_nodes, in1, emb, out):
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    mod = FullyConnectedTensorProduct(in1, f"{emb}x0e", out).to(dev)
    x = torch.randn(n_nodes, mod.irreps_in1.dim, device=dev, requires_grad=True)
    a = torch.randn(n_nodes, emb, device=dev, requires_grad=True)
    y = mod(x, a); g = torch.randn_like(y)
    flops = 2 * n_nodes * sum(ins[1] * ins[3] * ins[5] * ins[6] for ins in mod.spec.instructions)
    us = timeit(lambda: mod(x, a))
    print(f"fctp fwd {in1} -> {out}: {us:9.1f} us {flops/us/1e6:7.1f} TF/s")
    def bwd():
        y = mod(x, a)
        torch.autograd.grad(y, [x, a, mod.weight], g)
    us2 = timeit(bwd)
    print(f"fctp fwd+bwd: {us2:9.1f} us {4*flops/us2/1e6:7.1f} TF/s (bwd alone {us2-us:9.1f} us {3*flops/(us2-us)/1e6:7.1f} TF/s)")

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        plain(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "nn")
        sys.exit(0)
    which = sys.argv[1:] or ["plain", "fctp"]
    if "plain" in which:
        plain(4096, 4096, 4096)
        plain(4096, 4096, 4096, "nt")
        plain(4096, 4096, 4096, "tn")
        plain(59346, 1728, 64)
        plain(59346, 64, 1728, "nt")
        plain(64, 1728, 59346, "tn")
        plain(59346, 64, 64)
        plain(2048, 64, 4096)
        plain(10240, 64, 4096)
    if "fctp" in which:
        fctp(2048, "64x0e+64x1o+64x1e+64x2o+64x2e", 64, "64x0o+320x0e+64x1o+64x1e+64x2o+64x2e")
