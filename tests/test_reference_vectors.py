"""The reference's own tests on the rows next to the hot path, with their LITERAL inputs and expected
values (the reference's test files hold no numeric vector for the convolution itself), and
an independent check of the spherical-harmonic polynomials against scipy.

None of this lifts "parity unpinned" (DESIGN.md §2): a per-path sign convention of e3nn's TensorProduct
is still only restated, not observed.  It does pin: the species lookup, the masked selection, the
Cartesian round trip, the legacy constructor, and that the l <= 3 real spherical harmonics of SURVEY
App. A.2 are the standard ones in e3nn's axis convention (y polar; m = -l..l)."""
import math

import numpy as np
import pytest
import torch


def test_reference_atomic_number_to_index():
    """/root/reference/tests/nn/test_embedding.py:6-13, same inputs, same expected indices."""
    from matten_amd.nn.embedding import _AtomicNumberToIndex
    from oracle import eigenn_ref as R

    for cls in (_AtomicNumberToIndex, getattr(R, "_AtomicNumberToIndex", None)):
        if cls is None:
            continue
        n2i = cls([6, 1, 8])
        index = n2i(torch.tensor([6, 6, 8, 1, 8]))
        assert index.dtype == torch.long
        assert torch.equal(index, torch.tensor([1, 1, 2, 0, 2]))


def test_reference_nodewise_select():
    """/root/reference/tests/nn/test_nodewise.py:7-27: irreps_in values of None, an integer feature tensor."""
    from matten_amd import keys as DataKey
    from matten_amd.nn.nodewise import NodewiseSelect

    node_feats = DataKey.NODE_FEATURES
    mask_field = "node_masks"
    out_field = "selected_node_features"
    aws = NodewiseSelect(irreps_in={node_feats: None, mask_field: None}, field=node_feats, out_field=out_field, mask_field=mask_field)
    n_atoms = 5
    data = {
        node_feats: torch.arange(n_atoms * 2).reshape(n_atoms, 2),
        mask_field: torch.tensor([True, False, True, True, False]),
    }
    out = aws(data)
    assert torch.allclose(out[out_field], data[node_feats][data[mask_field]])
    assert torch.equal(out[out_field], torch.tensor([[0, 1], [4, 5], [6, 7]]))


def test_reference_irreps_to_cartesian():
    """/root/reference/tests/nn/test_readout.py:7-22."""
    from matten_amd import keys as DataKey
    from matten_amd.nn.readout import IrrepsToCartesianTensor

    i2c = IrrepsToCartesianTensor(irreps_in={DataKey.NODE_FEATURES: "0e + 2e"}, formula="ij=ji", out_field="my_out")
    t = torch.arange(9).to(torch.float).view(3, 3)
    y = i2c.ct.from_cartesian(t)
    z = i2c({DataKey.NODE_FEATURES: y})["my_out"]
    assert torch.allclose(z, (t + t.T) / 2, atol=1e-5)


def test_reference_legacy_constructor():
    """/root/reference/tests/nn/test_convolution.py:8-21: constructing PointConv with these irreps must not raise."""
    from matten_amd import keys as DataKey
    from matten_amd.nn.point_conv import PointConv

    conv = PointConv(
        irreps_in={
            DataKey.NODE_FEATURES: "4x1e+2x0e",
            DataKey.NODE_ATTRS: "4x1e+2x0e",
            DataKey.EDGE_ATTRS: "4x1e+2x0e",
            DataKey.EDGE_EMBEDDING: "4x0e",
        },
        irreps_out={DataKey.NODE_FEATURES: "8x1e+4x0e"},
    )
    assert str(conv.irreps_out[DataKey.NODE_FEATURES]) == "8x1e+4x0e"


def _real_sh_scipy(l, xyz):
    """Standard real spherical harmonics S_lm (m = -l..l; m < 0: sqrt2 (-1)^m Im Y_l^|m|, m > 0: sqrt2 (-1)^m Re Y_l^m),
    orthonormal on the sphere, from scipy's complex ones, evaluated in e3nn's axis convention: e3nn's (x, y, z)
    are the standard (Y, Z, X), i.e. y is the polar axis."""
    from scipy.special import sph_harm_y

    X, Y, Z = xyz[:, 2], xyz[:, 0], xyz[:, 1]
    theta = np.arccos(np.clip(Z, -1.0, 1.0))
    phi = np.arctan2(Y, X)
    cols = []
    for m in range(-l, l + 1):
        c = sph_harm_y(l, abs(m), theta, phi)
        if m < 0:
            cols.append(math.sqrt(2.0) * (-1) ** m * c.imag)
        elif m == 0:
            cols.append(c.real)
        else:
            cols.append(math.sqrt(2.0) * (-1) ** m * c.real)
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("l", [0, 1, 2, 3])
def test_spherical_harmonics_against_scipy(l):
    """The polynomials of SURVEY App. A.2 (oracle) and of matten_amd.so3 (what the product's tables are built
    from) equal scipy's spherical harmonics times sqrt(4 pi) ('component' normalisation), sign for sign."""
    from oracle import e3ref

    rng = np.random.default_rng(5 + l)
    v = rng.normal(size=(200, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    ref = math.sqrt(4.0 * math.pi) * _real_sh_scipy(l, v)
    irreps = "+".join(f"{k}{'e' if k % 2 == 0 else 'o'}" for k in range(l + 1))
    got = e3ref.spherical_harmonics(irreps, torch.as_tensor(v), True, "component").numpy()[:, l * l : (l + 1) * (l + 1)]
    assert np.abs(got - ref).max() < 1e-12
    # the other two normalisations e3nn offers: 'integral' (orthonormal on the sphere) and 'norm' (unit vector per l)
    got_i = e3ref.spherical_harmonics(irreps, torch.as_tensor(v), True, "integral").numpy()[:, l * l : (l + 1) * (l + 1)]
    assert np.abs(got_i - ref / math.sqrt(4.0 * math.pi)).max() < 1e-12
    got_n = e3ref.spherical_harmonics(irreps, torch.as_tensor(v), True, "norm").numpy()[:, l * l : (l + 1) * (l + 1)]
    assert np.abs(got_n - ref / math.sqrt(2 * l + 1)).max() < 1e-12
