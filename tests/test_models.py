"""The two target backbones end to end (SURVEY.md §8 rows f1/f3 and the "Python caller" row of §8c):
species embedding -> edge encoders -> 3 conv layers -> node-wise tail, built by factories that
mirror the reference's, on the reference's own 20 matbench structures (configs[0])."""
import numpy as np
import pytest
import torch

from tests import models as M
from tests.parity import rel_err

TOL = 1e-5


@pytest.fixture(scope="module")
def dataset():
    return M.load_dataset()


# ------------------------------------------------------------------------------------ CPU
def test_energy_model_structure(dataset):
    """Layer names, parameter count (71 680 + 120 192 + 142 208 per conv layer,
    + embedding + Bessel weights + the two read-out linears) and state_dict keys of the product
    equal the oracle's, which follow nequip_energy_model.py:63-202."""
    graphs, targets, hp = dataset
    assert len(graphs) == 20 and len(hp["allowed_species"]) == 33
    o, p = M.build("oracle", "energy", hp), M.build("hip", "energy", hp)
    names = ["one_hot", "spharm_edges", "radial_basis", "layer0_convnet", "layer1_convnet", "layer2_convnet",
             "conv_to_output_hidden", "output_hidden_to_scalar", "total_energy_sum"]
    assert [n for n, _ in o.named_children()] == names == [n for n, _ in p.named_children()]
    n_params = sum(q.numel() for q in p.parameters())
    assert n_params == 71680 + 120192 + 142208 + 33 * 32 + 8 + 32 * 16 + 16 == sum(q.numel() for q in o.parameters())
    assert list(o.state_dict()) == list(p.state_dict())
    assert {k: tuple(v.shape) for k, v in o.state_dict().items()} == {k: tuple(v.shape) for k, v in p.state_dict().items()}
    p.load_state_dict(o.state_dict(), strict=True)
    for k in ("total_energy", "atomic_energy", "node_features"):
        assert str(o.irreps_out[k]) == str(p.irreps_out[k])
    assert "one_hot.atomic_number_to_index._Z_to_index" in p.state_dict()  # the reference's buffer names


def test_atomic_tensor_model_structure(dataset):
    _, _, hp = dataset
    o, p = M.build("oracle", "tensor", hp), M.build("hip", "tensor", hp)
    assert [n for n, _ in p.named_children()][-4:] == ["conv_to_output_hidden", "output_hidden_to_tensor", "output_irreps_tensor", "output_cartesian_tensor"]
    assert list(o.state_dict()) == list(p.state_dict())
    assert str(p.irreps_out["atomic_output"]) == "1x0e+1x2e" == str(o.irreps_out["atomic_output"])
    bad = dict(M.MINIMAL_ATOMIC, output_format="spherical")
    from matten_amd import model_factory

    with pytest.raises(ValueError, match="output_format"):
        model_factory.create_atomic_tensor_model(bad, hp)


def test_irreps_to_cartesian_round_trip():
    """The reference's own test of this module (tests/nn/test_readout.py:7-22): from_cartesian followed by
    the module returns the symmetric part.  Plus: rows of the basis orthonormal, product == oracle."""
    from matten_amd import keys as K
    from matten_amd.nn import CartesianTensor, IrrepsToCartesianTensor
    from oracle import eigenn_ref as R

    i2c = IrrepsToCartesianTensor(irreps_in={K.NODE_FEATURES: "0e + 2e"}, formula="ij=ji", out_field="my_out")
    t = torch.arange(9).to(torch.float).view(3, 3)
    y = i2c.ct.from_cartesian(t)
    z = i2c({K.NODE_FEATURES: y})["my_out"]
    assert torch.allclose(z, (t + t.T) / 2, atol=1e-5)
    for formula, dim in (("ij=ji", 6), ("ij=-ji", 3), ("ij=ij", 9)):
        ct = CartesianTensor(formula)
        q = ct.change_of_basis.reshape(dim, 9)
        assert ct.dim == dim and torch.allclose(q @ q.T, torch.eye(dim, dtype=q.dtype), atol=1e-12)
        assert torch.allclose(ct.change_of_basis, R.cartesian_basis(formula)[0], atol=1e-12)
    with pytest.raises(AssertionError):
        IrrepsToCartesianTensor(irreps_in={K.NODE_FEATURES: "0e + 1e"}, formula="ij=ji")
    # an l=2 part transforms as a symmetric traceless matrix: Q(D y) = R Q(y) R^T
    from oracle import e3ref

    g = torch.Generator().manual_seed(3)
    A = torch.randn(3, 3, generator=g, dtype=torch.float64)
    Rm, _ = torch.linalg.qr(A)
    Rm = Rm * torch.sign(torch.linalg.det(Rm))
    ct = CartesianTensor("ij=ji")
    yv = torch.randn(6, generator=g, dtype=torch.float64)
    D = e3ref.irreps_D(e3ref.Irreps("0e+2e"), Rm)
    assert torch.allclose(ct.to_cartesian(D @ yv), Rm @ ct.to_cartesian(yv) @ Rm.T, atol=1e-10)


def test_species_embedding_errors():
    from matten_amd.nn import SpeciesEmbedding

    with pytest.raises(ValueError):
        SpeciesEmbedding(num_species=3, allowed_species=[1, 2, 3])
    m = SpeciesEmbedding(embedding_dim=4, allowed_species=[8, 1, 14])
    assert torch.equal(m.atomic_number_to_index(torch.tensor([1, 14, 8, 8])), torch.tensor([0, 2, 1, 1]))
    with pytest.raises(RuntimeError):
        m.atomic_number_to_index(torch.tensor([1, 99]))
    with pytest.raises(RuntimeError):
        m.atomic_number_to_index(torch.tensor([1, 6]))
    with pytest.raises(ValueError):
        m({})


def test_sequential_checks_irreps():
    from collections import OrderedDict

    from matten_amd.nn import NodewiseLinear, Sequential, create_sequential_module

    a = NodewiseLinear({"node_features": "4x0e"}, irreps_out="8x0e")
    b = NodewiseLinear({"node_features": "4x0e"}, irreps_out="2x0e")
    with pytest.raises(ValueError, match="incompatible"):
        Sequential(a, b)
    with pytest.raises(RuntimeError, match="Failed instantiate"):
        create_sequential_module(OrderedDict(x=(NodewiseLinear, {"field": "missing"})), irreps_in={"node_features": "4x0e"})
    with pytest.raises(ValueError):
        create_sequential_module(OrderedDict(x=(NodewiseLinear, {"irreps_in": {}})), irreps_in={"node_features": "4x0e"})
    seq = create_sequential_module(OrderedDict(x=(NodewiseLinear, {"irreps_out": "8x0e"}), y=(NodewiseLinear, {"irreps_out": "1x0e", "out_field": "e"})),
                                   irreps_in={"node_features": "4x0e"})
    assert str(seq.irreps_out["e"]) == "1x0e" and str(seq.irreps_out["node_features"]) == "8x0e"


def test_oracle_training_trajectory_is_stable(dataset):
    """2 epochs of minimal.yaml on the 20 structures with the oracle, fp32 against fp64: the loss
    trajectory the GPU test compares against is not chaotic at this length."""
    graphs, targets, hp = dataset
    torch.manual_seed(35)
    o = M.build("oracle", "energy", hp)
    state = {k: v.clone() for k, v in o.state_dict().items()}
    l64 = M.train_trajectory(o, graphs, targets, "cpu", torch.float64)
    o32 = M.build("oracle", "energy", hp)
    o32.load_state_dict(state)
    l32 = M.train_trajectory(o32, graphs, targets, "cpu", torch.float32)
    assert len(l64) == 4 and np.allclose(l32, l64, rtol=1e-5)
    assert l64[2] < l64[0] and l64[3] < l64[1]  # second epoch improves on the first, batch by batch


# ------------------------------------------------------------------------------------ GPU
def _grads(model, out_key, batch, weight):
    out = model(batch)
    val = out[out_key]
    loss = (val * weight.to(val)).sum()
    params = list(model.parameters())
    grads = torch.autograd.grad(loss, [batch["pos"]] + params, allow_unused=True)
    return val.detach(), grads


@pytest.mark.gpu
@pytest.mark.parametrize("ids", [list(range(10)), list(range(10, 20)), [4]])
def test_energy_model_vs_oracle(device, dataset, ids):
    """total_energy, forces and every parameter gradient of the minimal.yaml model on real structures
    (33 species, 5-36 atoms, triclinic cells), 1e-5 relative to the fp64 oracle."""
    graphs, targets, hp = dataset
    torch.manual_seed(35)
    o = M.build("oracle", "energy", hp).double()
    p = M.build("hip", "energy", hp)
    p.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in o.state_dict().items()}, strict=True)
    p = p.to(device)
    bo, bp = M.make_batch(graphs, ids, "cpu", torch.float64), M.make_batch(graphs, ids, device, torch.float32)
    bo["pos"].requires_grad_(True)
    bp["pos"].requires_grad_(True)
    w = torch.randn(len(ids), 1, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    eo, go = _grads(o, "total_energy", bo, w)
    ep, gp = _grads(p, "total_energy", bp, w.to(device))
    assert ep.shape == eo.shape == (len(ids), 1)
    assert rel_err(ep, eo) < TOL
    names = ["pos"] + [n for n, _ in o.named_parameters()]
    for n, a, b in zip(names, gp, go):
        assert (a is None) == (b is None), n
        if b is not None:
            assert rel_err(a, b) < TOL, n


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["cartesian", "irreps"])
def test_atomic_tensor_model_vs_oracle(device, dataset, fmt):
    """minimal_atomic.yaml's backbone: per-atom symmetric tensors on masked atoms."""
    graphs, _, hp = dataset
    torch.manual_seed(36)
    hparams = dict(M.MINIMAL_ATOMIC, output_format=fmt, output_formula="ij=ji" if fmt == "cartesian" else "0e + 2e")
    from matten_amd import model_factory
    from oracle import eigenn_ref as R

    o = R.create_atomic_tensor_model(dict(hparams), hp).double()
    p = model_factory.create_atomic_tensor_model(dict(hparams), hp)
    p.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in o.state_dict().items()}, strict=True)
    p = p.to(device)
    ids = [0, 3, 7, 13, 19]
    bo, bp = M.make_batch(graphs, ids, "cpu", torch.float64), M.make_batch(graphs, ids, device, torch.float32)
    n = bo["pos"].shape[0]
    mask = torch.rand(n, generator=torch.Generator().manual_seed(2)) < 0.2
    bo["node_masks"], bp["node_masks"] = mask, mask.to(device)
    bo["pos"].requires_grad_(True)
    bp["pos"].requires_grad_(True)
    shape = (int(mask.sum()), 3, 3) if fmt == "cartesian" else (int(mask.sum()), 6)
    w = torch.randn(shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    to, go = _grads(o, "tensor_output", bo, w)
    tp, gp = _grads(p, "tensor_output", bp, w.to(device))
    assert tuple(tp.shape) == shape == tuple(to.shape)
    assert rel_err(tp, to) < TOL
    if fmt == "cartesian":
        assert torch.equal(tp, tp.transpose(1, 2)) or rel_err(tp, tp.transpose(1, 2)) < 1e-6
    for nme, a, b in zip(["pos"] + [k for k, _ in o.named_parameters()], gp, go):
        assert (a is None) == (b is None), nme
        if b is not None:
            assert rel_err(a, b) < TOL, nme


@pytest.mark.gpu
@pytest.mark.parametrize("reduce", ["sum", "mean", "min", "max"])
def test_nodewise_reduce_vs_oracle(device, reduce):
    from matten_amd.nn import NodewiseReduce
    from oracle import eigenn_ref as R

    g = torch.Generator().manual_seed(5)
    x = torch.randn(50, 7, generator=g, dtype=torch.float64, requires_grad=True)
    batch = torch.sort(torch.randint(0, 6, (50,), generator=g)).values
    batch[-1] = 5
    o = R.NodewiseReduce({"f": "7x0e"}, field="f", reduce=reduce)
    p = NodewiseReduce({"f": "7x0e"}, field="f", reduce=reduce)
    assert p.out_field == f"{reduce}_f"
    xo = o({"f": x, "batch": batch})[f"{reduce}_f"]
    xg = x.detach().float().to(device).requires_grad_(True)
    xp = p({"f": xg, "batch": batch.to(device)})[f"{reduce}_f"]
    assert rel_err(xp, xo) < TOL
    w = torch.randn(xo.shape, generator=g, dtype=torch.float64)
    (go,) = torch.autograd.grad((xo * w).sum(), [x])
    (gp,) = torch.autograd.grad((xp * w.to(xp)).sum(), [xg])
    assert rel_err(gp, go) < TOL


@pytest.mark.gpu
def test_training_trajectory_vs_oracle(device, dataset):
    """configs[0] as the reference runs it (minimal.yaml: seed 35, batch 10, L1Loss on total_energy vs
    y["n"], Adam lr 1e-3, 2 epochs over the 20 matbench structures; fixed batch order instead of the
    shuffle): the 4 training losses of the HIP path against the fp64 oracle's.  Tolerance 1e-4 relative:
    Adam's first steps are sign-like, so fp32 rounding in tiny gradients moves parameters by O(lr);
    the fp32 oracle itself stays within 1e-6 of the fp64 one on this run (CPU test above)."""
    graphs, targets, hp = dataset
    torch.manual_seed(35)
    o = M.build("oracle", "energy", hp)
    state = {k: v.clone() for k, v in o.state_dict().items()}
    ref = M.train_trajectory(o, graphs, targets, "cpu", torch.float64)
    p = M.build("hip", "energy", hp)
    p.load_state_dict(state, strict=True)
    got = M.train_trajectory(p, graphs, targets, device, torch.float32)
    print("oracle fp64:", ref, "\nhip  fp32:", got)
    assert np.allclose(got, ref, rtol=1e-4)
    # and the trained parameters agree
    so = o.state_dict()
    for k, v in p.state_dict().items():
        if v.is_floating_point():
            assert rel_err(v, so[k]) < 1e-3, k


def test_lsdi_fixture_unpacks_like_the_reference():
    """20 entries (24x4, 9x4, 60x4, 28x8 atoms), 1 tensor each except 3 for the 60-atom
    ones -> 28 data points, each with exactly one selected site and a symmetric target."""
    graphs, masks, targets, hp = M.load_lsdi()
    assert len(graphs) == 28 and hp["allowed_species"] == (8, 12, 14, 20)
    assert sorted({len(g["pos"]) for g in graphs}) == [9, 24, 28, 60]
    assert all(int(m.sum()) == 1 for m in masks)
    assert all(torch.equal(t, t.transpose(1, 2)) and t.shape == (1, 3, 3) for t in targets)


@pytest.mark.gpu
def test_atomic_tensor_training_trajectory_vs_oracle(device):
    """configs[1] (minimal_atomic.yaml on the LSDI NMR fixture, 1 GPU): 2 epochs x 3 batches of L1Loss on the
    selected sites' symmetric tensors, HIP path against the fp64 oracle, same tolerance as the energy run."""
    graphs, masks, targets, hp = M.load_lsdi()
    torch.manual_seed(35)
    o = M.build("oracle", "tensor", hp)
    state = {k: v.clone() for k, v in o.state_dict().items()}
    ref = M.tensor_train_trajectory(o, graphs, masks, targets, "cpu", torch.float64)
    p = M.build("hip", "tensor", hp)
    p.load_state_dict(state, strict=True)
    got = M.tensor_train_trajectory(p, graphs, masks, targets, device, torch.float32)
    print("oracle fp64:", ref, "\nhip  fp32:", got)
    assert len(ref) == 6 and np.allclose(got, ref, rtol=1e-4)
