"""Generates tests/golden/stack_<config>.npz: seeded inputs and the ORACLE's fp64 outputs for
the hot path end to end (tests/parity.py: embedding -> SH -> radial -> conv layers -> energy,
forces, selected parameter gradients).  The reference cannot run here (e3nn absent) and
holds no vectors for this path, so these pin the oracle against drift and give the HIP path
a committed target ("parity unpinned" with respect to real e3nn; see oracle/__init__.py).

Weights are not stored: they are regenerated from torch.manual_seed(35) (CPU randn is
reproducible), and the first values are stored as a guard.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

GOLDEN_CONFIGS = ["smoke", "minimal", "lmax3"]
GRAD_KEYS = [
    "grad:embedding.weight",
    "grad:radial_basis.basis.bessel_weights",
    "grad:convs.0.message_fn.linear_1.weight",
    "grad:convs.1.message_fn.tp.weight_nn.layer0.weight",
    "grad:convs.1.update_fn.linear_2.weight",
]


def main():
    from tests.parity import CONFIGS, Stack, make_batch, run_stack

    for name in GOLDEN_CONFIGS:
        cfg = CONFIGS[name]
        torch.manual_seed(35)
        batch = make_batch(cfg)
        model = Stack("oracle", cfg)
        guard = model.convs[0].message_fn.linear_1.weight.detach()[:8].double().numpy().copy()
        res = run_stack(model, batch, "cpu", torch.float64)
        out = {
            "pos": batch["pos"].numpy(),
            "cell": batch["cell"].numpy(),
            "edge_index": batch["edge_index"].numpy(),
            "edge_cell_shift": batch["edge_cell_shift"].numpy(),
            "atomic_numbers": batch["atomic_numbers"].numpy(),
            "batch": batch["batch"].numpy(),
            "weight_guard": guard,
            "feats": res["feats"].numpy(),
            "energy": res["energy"].numpy(),
            "forces": res["forces"].numpy(),
            "edge_attrs": res["edge_attrs"].numpy(),
            "edge_embedding": res["edge_embedding"].numpy(),
        }
        for k in GRAD_KEYS:
            g = res[k].numpy().reshape(-1)
            out[k] = g[:: max(1, len(g) // 4096)]  # a strided sample keeps the fixture small
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"stack_{name}.npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB", "E =", out["edge_index"].shape[1])


if __name__ == "__main__":
    main()
