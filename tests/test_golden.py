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


# ------------------------------------------------------------------ legacy PointConv (pointconv_legacy.npz)
def _pointconv(impl):
    from tests.golden import make_pointconv_golden as G

    if impl == "oracle":
        from oracle import eigenn_ref as M
    else:
        import matten_amd.nn as M
    torch.manual_seed(35)
    return M.PointConv(G.IRREPS_IN, G.IRREPS_OUT, **G.KWARGS), G


def _compare_pointconv(res, tol):
    z = np.load(os.path.join(HERE, "golden", "pointconv_legacy.npz"), allow_pickle=False)
    keys = [k for k in z.files if k != "weight_guard"]
    assert sorted(keys) == sorted(res)
    for k in keys:
        assert rel_err(res[k], torch.from_numpy(z[k])) <= tol, k
    return z


def test_oracle_reproduces_pointconv_golden():
    layer, G = _pointconv("oracle")
    res = G.run(layer.double(), *G.inputs())
    z = _compare_pointconv(res, 1e-12)
    assert np.allclose(layer.linear_1.weight.detach()[:8].numpy(), z["weight_guard"])


@pytest.mark.gpu
def test_hip_reproduces_pointconv_golden(device):
    """The legacy single-module layer of the reference's constructor test, through the C ABI."""
    ref, G = _pointconv("oracle")
    layer, _ = _pointconv("hip")
    layer.load_state_dict(ref.state_dict(), strict=True)
    layer = layer.to(device)
    args = [t.to(device) if not t.is_floating_point() else t.float().to(device) for t in G.inputs()]
    _compare_pointconv(G.run(layer, *args), 1e-5)


# ------------------------------------------------------------------ the reference test's literal irreps (pointconv_literal.npz)
def _pointconv_literal(impl):
    from tests.golden import make_pointconv_literal_golden as G

    if impl == "oracle":
        from oracle import eigenn_ref as M
    else:
        import matten_amd.nn as M
    torch.manual_seed(35)
    return M.PointConv(G.IRREPS_IN, G.IRREPS_OUT, **G.KWARGS), G


def _compare_literal(res, tol):
    z = np.load(os.path.join(HERE, "golden", "pointconv_literal.npz"), allow_pickle=False)
    keys = [k for k in z.files if k != "weight_guard"]
    assert sorted(keys) == sorted(res)
    for k in keys:
        assert rel_err(res[k], torch.from_numpy(z[k])) <= tol, k
    return z


def test_oracle_reproduces_pointconv_literal_golden():
    layer, G = _pointconv_literal("oracle")
    res = G.run(layer.double(), *G.inputs())
    z = _compare_literal(res, 1e-12)
    assert np.allclose(layer.linear_1.weight.detach()[:8].numpy(), z["weight_guard"])


@pytest.mark.gpu
def test_hip_reproduces_pointconv_literal_golden(device):
    """What the reference's own test constructs (tests/nn/test_convolution.py:8-21: edge attributes 4x1e+2x0e, i.e.
    second-input blocks of multiplicity 4 and 2 in the 'uvu' product, and NON-scalar node attributes in the self
    connection) runs forward and backward on the HIP kernels and matches the oracle's fixture."""
    ref, G = _pointconv_literal("oracle")
    layer, _ = _pointconv_literal("hip")
    assert layer.tp.tp._expanded and layer.self_connection._general
    layer.load_state_dict(ref.state_dict(), strict=True)
    layer = layer.to(device)
    args = [t.to(device) if not t.is_floating_point() else t.float().to(device) for t in G.inputs()]
    _compare_literal(G.run(layer, *args), 1e-5)


@pytest.mark.gpu
def test_hip_refuses_l_above_3(device):
    """What the kernels do not cover (l > 3) constructs and raises NotImplementedError when called."""
    import matten_amd.nn as H

    conv = H.PointConv(
        irreps_in={"node_features": "2x4e+2x0e", "node_attrs": "3x0e", "edge_attrs": "0e+1o", "edge_embedding": "4x0e"},
        irreps_out={"node_features": "2x4e+4x0e"},
    ).to(device)
    data = {"node_features": torch.randn(5, 20, device=device), "node_attrs": torch.randn(5, 3, device=device),
            "edge_attrs": torch.randn(9, 4, device=device), "edge_embedding": torch.randn(9, 4, device=device),
            "edge_index": torch.randint(0, 5, (2, 9), device=device)}
    with pytest.raises(NotImplementedError):
        conv(data)
