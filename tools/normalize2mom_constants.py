"""Regenerates the normalize2mom constants baked into matten_amd/nn/e3.py (e3nn.math.normalize2mom:
cst = E[act(z)^2]^(-1/2), z = torch.randn(1_000_000, generator=manual_seed(0), dtype=float64) on CPU)."""
import math

import torch


def ssp(x):
    return torch.nn.functional.softplus(x) - math.log(2.0)


ACTS = {"ssp": ssp, "silu": torch.nn.functional.silu, "tanh": torch.tanh, "abs": torch.abs, "sigmoid": torch.sigmoid}

if __name__ == "__main__":
    for name, f in ACTS.items():
        z = torch.randn(1_000_000, generator=torch.Generator(device="cpu").manual_seed(0), dtype=torch.float64)
        print(f'    "{name}": {f(z).pow(2).mean().pow(-0.5).item()!r},')
