"""Generates tests/golden/pointconv_literal.npz: the legacy `PointConv` layer (eigenn/nn/point_conv.py:351-491) on
EXACTLY the irreps of the reference's own constructor test (tests/nn/test_convolution.py:8-21): node features, node
attributes and edge attributes all 4x1e+2x0e (so the 'uvu' product sees second-input blocks of multiplicity 4 and 2,
and the self connection a non-scalar second input), EDGE_EMBEDDING 4x0e, output 8x1e+4x0e, default kwargs - evaluated
by the ORACLE in fp64 on a seeded random graph.  Inputs, outputs and gradients; weights from torch.manual_seed(35).
"Parity unpinned" with respect to real e3nn (oracle/__init__.py).

    python tests/golden/make_pointconv_literal_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

IRREPS_IN = {"node_features": "4x1e+2x0e", "node_attrs": "4x1e+2x0e", "edge_attrs": "4x1e+2x0e", "edge_embedding": "4x0e"}
IRREPS_OUT = {"node_features": "8x1e+4x0e"}
KWARGS = dict()
N, E = 31, 211


def inputs():
    g = torch.Generator().manual_seed(92)
    ei = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, 14, generator=g, dtype=torch.float64)
    a = torch.randn(N, 14, generator=g, dtype=torch.float64)
    y = torch.randn(E, 14, generator=g, dtype=torch.float64)
    emb = torch.randn(E, 4, generator=g, dtype=torch.float64)
    go = torch.randn(N, 28, generator=g, dtype=torch.float64)
    return ei, x, a, y, emb, go


def run(layer, ei, x, a, y, emb, go):
    x, a, y, emb = (t.clone().requires_grad_(True) for t in (x, a, y, emb))
    out = layer({"node_features": x, "node_attrs": a, "edge_attrs": y, "edge_embedding": emb, "edge_index": ei})["node_features"]
    params = list(layer.parameters())
    grads = torch.autograd.grad(out, [x, a, y, emb] + params, go.to(out))
    res = {"out": out.detach(), "dx": grads[0], "da": grads[1], "dy": grads[2], "demb": grads[3]}
    for (n, _), g in zip(layer.named_parameters(), grads[4:]):
        res["grad:" + n] = g
    return res


def main():
    from oracle import eigenn_ref as R

    torch.manual_seed(35)
    layer = R.PointConv(IRREPS_IN, IRREPS_OUT, **KWARGS).double()
    res = run(layer, *inputs())
    out = {k: v.numpy() for k, v in res.items()}
    out["weight_guard"] = layer.linear_1.weight.detach()[:8].numpy().copy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pointconv_literal.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", sorted(out))


if __name__ == "__main__":
    main()
