#!/usr/bin/env python
"""SURVEY.md §7 step 6 "measure both": forward of (radial network + convolution) with the per-edge weights
w [E, W] written by the network and read by the convolution (shipped), against the convolution forming them on
the fly from the last hidden activations (tfc_conv_fwd_fused_m; w never exists).  Config-3 and config-4 last-layer
shapes; times from the library's HIP events; algorithmic bytes with the reduced formula for the fused call.
usage: bench_fused_fwd.py [c3|c4]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matten_amd import ops, keys as K, _lib
from matten_amd.nn.utils import UVUTensorProduct
from matten_amd.data.crystal import synthetic_batch

dev = torch.device("cuda")
CASES = {
    "c3": dict(n=4, graphs=32, lmax=2, in1="64x0e+64x1o+64x1e+64x2o+64x2e", out="64x0o+320x0e+64x1o+64x1e+64x2o+64x2e", sh="0e+1o+2e"),
    "c4": dict(n=6, graphs=32, lmax=3, in1="128x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e",
               out="128x0o+640x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e", sh="0e+1o+2e+3o"),
}


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L = _lib.lib(); L.tfc_profile_reset(); L.tfc_profile_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    L.tfc_profile_enable(0)
    r = {k: tot / c * 1e3 for k, (c, tot) in _lib.profile_report().items()}
    L.tfc_profile_reset()
    return r


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    c = CASES[wl]
    batch, _ = synthetic_batch(c["n"], 2.5, c["graphs"], seed0=1000, r_cut=5.0)
    ei = batch[K.EDGE_INDEX].to(dev)
    N, E = batch[K.POSITIONS].shape[0], ei.shape[1]
    order = ops.EdgeOrdering(ei, N)
    torch.manual_seed(0)
    y = ops.spherical_harmonics(torch.randn(E, 3, device=dev), c["lmax"])
    tp = UVUTensorProduct(c["in1"], c["sh"], c["out"], mlp_input_size=8, mlp_hidden_size=64, mlp_num_hidden_layers=2).to(dev)
    core, net = tp.tp, tp.weight_nn
    plan = core.plan
    d_in, d_mid, W, S, H = core.irreps_in1.dim, core.irreps_out.dim, core.weight_numel, y.shape[1], 64
    x = torch.randn(N, d_in, device=dev)
    emb = torch.randn(E, 8, device=dev)
    ws = [p.detach() for p in net._weights()]
    L = _lib.lib()
    rows = order.by_dst
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out_f = torch.empty(N, d_mid, device=dev)

    with torch.no_grad():
        def shipped():
            w = net(emb)
            return core(x, y, w, order, 0.2)

        def fused():
            # hidden layers only (two small products with the activation in the epilogue), then the fused convolution
            h = emb
            for wt in ws[:-1]:
                o = torch.empty(E, wt.shape[1], device=dev)
                ops._gemm(E, wt.shape[1], wt.shape[0], ops._p(h), ops._p(wt), ops._p(o), a1_r=ops._lin(wt.shape[0]), a1_k=ops._lin(1),
                          b_k=ops._lin(wt.shape[1]), b_j=ops._lin(1), c_r=ops._lin(wt.shape[1]), c_j=ops._lin(1),
                          alpha=1.0 / math.sqrt(wt.shape[0]), epilogue=_lib.EPI_SSP, act_cst=net.act_cst, tag=b"gemm_mlp_hidden")
                h = o
            M = order.m_matrices(plan, y, rows)
            _lib.check(L.tfc_conv_fwd_fused_m(plan.handle(dev), P(x), P(M), P(h), P(ws[-1]), H, 1.0 / math.sqrt(H), P(rows[0]), P(rows[1]),
                                              P(rows[2]), N, E, 0.2, P(out_f), st))
            return out_f

        ref = shipped()
        got = fused().clone()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        a = timed(shipped)
        b = timed(fused)
    deg = E / N
    b_fwd = (4 * (d_in + S + W) + 16 + 4 * d_mid / deg) * E
    b_fused = (4 * (d_in + S + H) + 16 + 4 * d_mid / deg) * E  # the w row is gone, the h row is read instead
    b_mlp = 4 * (8 + 2 * H + W) * E  # what the network's forward writes (hidden activations + w) and reads (embedding)
    t_mlp, t_conv = a["mlp_fused_fwd"], a["tfc_conv_fwd_kernel"]
    t_hid, t_fconv = b["gemm_mlp_hidden"] * 2, b["tfc_conv_fwd_fused_kernel"]
    print(f"{wl} last layer: N={N} E={E} d_in={d_in} W={W} d_mid={d_mid}; fused vs shipped max-norm difference {err:.1e}")
    print(f"  shipped: radial network {t_mlp:8.1f} us + convolution {t_conv:8.1f} us = {t_mlp + t_conv:8.1f} us   "
          f"(conv {b_fwd / t_conv / 1e3:6.0f} GB/s algorithmic; network + conv {(b_fwd + b_mlp) / (t_mlp + t_conv) / 1e3:6.0f} GB/s)")
    print(f"  fused  : hidden layers  {t_hid:8.1f} us + convolution {t_fconv:8.1f} us = {t_hid + t_fconv:8.1f} us   "
          f"(conv {b_fused / t_fconv / 1e3:6.0f} GB/s with the reduced byte formula)")


if __name__ == "__main__":
    main()
