"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py from the oracle
in fp64).  CPU: the oracle still reproduces them (guards the restatement against drift).
GPU: the HIP path, through the C ABI, reproduces them to 1e-5 relative (fp32, max-norm)."""
import os

import numpy as np
import pytest
import torch

from tests.golden.make_golden import GOLDEN_CONFIGS, GRAD_KEYS
from tests.parity import CONFIGS, Stack, rel_err, run_stack

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    z = np.load(os.path.join(HERE, "golden", f"stack_{name}.npz"), allow_pickle=False)
    batch = {k: torch.from_numpy(z[k]) for k in ("pos", "cell", "edge_index", "edge_cell_shift", "atomic_numbers", "batch")}
    return z, batch


def compare(res, z, tol):
    worst = 0.0
    for k in ("feats", "energy", "forces", "edge_attrs", "edge_embedding"):
        worst = max(worst, rel_err(res[k], torch.from_numpy(z[k])))
    for k in GRAD_KEYS:
        g = res[k].reshape(-1)
        g = g[:: max(1, len(g) // 4096)]
        worst = max(worst, rel_err(g, torch.from_numpy(z[k])))
    assert worst <= tol, worst


@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_oracle_reproduces_golden(name):
    z, batch = load(name)
    torch.manual_seed(35)
    # the generator also draws the batch before the weights; inputs come from the fixture here,
    # the weights from the same seed
    from tests.parity import make_batch

    make_batch(CONFIGS[name])
    model = Stack("oracle", CONFIGS[name])
    assert np.allclose(model.convs[0].message_fn.linear_1.weight.detach()[:8].double().numpy(), z["weight_guard"])
    compare(run_stack(model, batch, "cpu", torch.float64), z, 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOLDEN_CONFIGS)
def test_hip_reproduces_golden(device, name):
    z, batch = load(name)
    torch.manual_seed(35)
    from tests.parity import make_batch

    make_batch(CONFIGS[name])
    ref = Stack("oracle", CONFIGS[name])
    assert np.allclose(ref.convs[0].message_fn.linear_1.weight.detach()[:8].double().numpy(), z["weight_guard"])
    model = Stack("hip", CONFIGS[name])
    model.load_state_dict(ref.state_dict(), strict=True)
    compare(run_stack(model, batch, str(device), torch.float32), z, 1e-5)
