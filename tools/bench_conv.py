#!/usr/bin/env python
"""Micro-benchmark of the fused conv kernels on the config-3 layer shapes (GPU only).
Prints per-layer forward / backward times and achieved algorithmic GB/s (SURVEY.md 8d formulas)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matten_amd import ops, keys as K, _lib
from matten_amd.nn.utils import UVUTensorProduct
from matten_amd.data.crystal import synthetic_batch

dev = torch.device("cuda")

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    L = _lib.lib(); L.tfc_profile_reset(); L.tfc_profile_enable(1)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    L.tfc_profile_enable(0)
    return {k: tot / c * 1e3 for k, (c, tot) in _lib.profile_report().items()}

LAYERS = {
    "c3": [("64x0e", "64x0o+320x0e+64x1o+64x1e+64x2o+64x2e"),
           ("64x0e+64x1o+64x2e", "64x0o+320x0e+64x1o+64x1e+64x2o+64x2e"),
           ("64x0e+64x1o+64x1e+64x2o+64x2e", "64x0o+320x0e+64x1o+64x1e+64x2o+64x2e")],
}

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    batch, _ = synthetic_batch(4, 2.5, 32, seed0=1000, r_cut=5.0)
    ei = batch[K.EDGE_INDEX].to(dev)
    N, E = batch[K.POSITIONS].shape[0], ei.shape[1]
    order = ops.EdgeOrdering(ei, N)
    v = torch.randn(E, 3, device=dev)
    y = ops.spherical_harmonics(v, 2)
    for li, (in1, out) in enumerate(LAYERS[wl]):
        tp = UVUTensorProduct(in1, "0e+1o+2e", out, mlp_input_size=8, mlp_hidden_size=64, mlp_num_hidden_layers=2).to(dev)
        core = tp.tp
        d_in, d_mid, W, S = core.irreps_in1.dim, core.irreps_out.dim, core.weight_numel, 9
        x = torch.randn(N, d_in, device=dev, requires_grad=True)
        w = torch.randn(E, W, device=dev, requires_grad=True)
        deg = E / N
        b_fwd = (4 * (d_in + S + W) + 16 + 4 * d_mid / deg) * E
        b_bwd = (4 * (d_mid + d_in + S + W) + 16 + 4 * (W + S) + 4 * d_in / deg) * E
        for need_dy in (True, False):
            yy = y.detach().requires_grad_(need_dy)
            out_t = core(x, yy, w, order, 0.2)
            g = torch.randn_like(out_t)
            def step():
                o = core(x, yy, w, order, 0.2)
                torch.autograd.grad(o, [x, w] + ([yy] if need_dy else []), g)
            r = timeit(step)
            f, b = r["tfc_conv_fwd_kernel"], r["tfc_conv_bwd_kernel"]
            print(f"L{li} d_in={d_in} d_mid={d_mid} W={W} dy={need_dy}: fwd {f:7.1f} us {b_fwd/f/1e3:7.0f} GB/s | bwd {b:7.1f} us {b_bwd/b/1e3:7.0f} GB/s | prepass {r.get('tfc_m_prepass_kernel', 0.0):6.1f} us")

if __name__ == "__main__":
    main()
