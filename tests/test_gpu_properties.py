"""-m gpu: size-independent properties of the hot path at BASELINE.json's FULL sizes (config 3:
2048 atoms / ~59 k edges; config 4: 6912 atoms / ~200 k edges, lmax 3, 128 channels; config 5: one
1728-atom cell / ~97 k edges), each through the whole 3-layer energy model, where the CPU oracle
would take minutes: invariances and covariances the domain guarantees, linearity, batch
independence, determinism, and bit-exact structure of the integer work."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C3 = dict(n=4, a=2.5, graphs=32, emb=64, sh="0e + 1o + 2e", irreps="64x0o+64x0e+64x1o+64x1e+64x2o+64x2e")
C4 = dict(n=6, a=2.5, graphs=32, emb=128, sh="0e + 1o + 2e + 3o", irreps="128x0o+128x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e")
C5 = dict(n=12, a=2.1, graphs=1, emb=64, sh="0e + 1o + 2e", irreps="64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", avg=56.0)
FULL = {"c3": C3, "c4": C4, "c5": C5}


def _hparams(cfg, layers=3):
    return dict(
        species_embedding_dim=cfg["emb"], irreps_edge_sh=cfg["sh"], num_radial_basis=8, radial_basis_r_cut=5.0, num_layers=layers,
        invariant_layers=2, invariant_neurons=64, average_num_neighbors=cfg.get("avg", 28.0), conv_layer_irreps=cfg["irreps"],
        nonlinearity_type="gate", use_sc=True, resnet=True, conv_to_output_hidden_irreps_out="16x0e", reduce="sum",
    )


@pytest.fixture(scope="module", params=["c3", "c4", "c5"])
def c3(request, device):
    """(model, batch, graphs, atoms per graph) of one full-size configuration; the fixture keeps its first name."""
    from matten_amd import model_factory
    from matten_amd.data.crystal import synthetic_batch

    cfg = FULL[request.param]
    batch, graphs = synthetic_batch(cfg["n"], cfg["a"], cfg["graphs"], seed0=1000, r_cut=5.0)
    torch.manual_seed(35)
    model = model_factory.create_energy_model(_hparams(cfg), {"allowed_species": (1, 6, 8, 14)}).to(device)
    yield model, {k: v.to(device) for k, v in batch.items()}, graphs
    del model
    torch.cuda.empty_cache()


def _energy_forces(model, batch, grads=False):
    b = dict(batch)
    b["pos"] = batch["pos"].detach().clone().requires_grad_(True)
    e = model(b)["total_energy"].reshape(-1)
    params = list(model.parameters()) if grads else []
    g = torch.autograd.grad(e.sum(), [b["pos"]] + params)
    return e.detach(), -g[0], [t.detach() for t in g[1:]]


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


def _rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    return q * torch.sign(torch.linalg.det(q))


def test_full_size_rotation_and_inversion(c3, device):
    """E(R x) = E(x), F(R x) = R F(x) for a proper rotation and for the inversion (cells rotated
    along; cell shifts are integers and stay)."""
    model, batch, _ = c3
    e0, f0, _ = _energy_forces(model, batch)
    assert batch["pos"].shape[0] in (2048, 6912, 1728) and e0.shape[0] in (32, 1)
    for Rm in (_rotation(5), -torch.eye(3, dtype=torch.float64), -_rotation(6)):
        Rf = Rm.to(device=device, dtype=torch.float32)
        b = dict(batch)
        b["pos"] = batch["pos"] @ Rf.T
        b["cell"] = batch["cell"] @ Rf.T
        e1, f1, _ = _energy_forces(model, b)
        assert _rel(e1, e0) < 1e-5
        assert _rel(f1, f0 @ Rf.T) < 2e-5  # two fp32 evaluations, each within 1e-5 of the exact value


def test_full_size_edge_and_graph_permutation(c3, device):
    """Neither the order of the edge list nor the order of the graphs in the batch matters (the
    kernels regroup edges themselves); sums are re-ordered, so equality is to rounding."""
    from matten_amd.data.crystal import collate

    model, batch, graphs = c3
    e0, f0, _ = _energy_forces(model, batch)
    perm = torch.randperm(batch["edge_index"].shape[1], generator=torch.Generator().manual_seed(1)).to(device)
    b = dict(batch)
    b["edge_index"] = batch["edge_index"][:, perm].contiguous()
    b["edge_cell_shift"] = batch["edge_cell_shift"][perm].contiguous()
    e1, f1, _ = _energy_forces(model, b)
    assert _rel(e1, e0) < 1e-5 and _rel(f1, f0) < 1e-5
    order = list(reversed(range(len(graphs))))
    b2 = {k: v.to(device) for k, v in collate([graphs[i] for i in order]).items()}
    e2, f2, _ = _energy_forces(model, b2)
    assert _rel(e2.flip(0), e0) < 1e-5
    n_g = len(graphs)
    n_at = batch["pos"].shape[0] // n_g
    assert _rel(f2.reshape(n_g, n_at, 3).flip(0).reshape(-1, 3), f0) < 1e-5


def test_full_size_batch_independence_and_translation(c3, device):
    """A graph's energy and forces do not depend on what else is in the batch, nor on a rigid translation."""
    from matten_amd.data.crystal import collate

    model, batch, graphs = c3
    e0, f0, _ = _energy_forces(model, batch)
    n_g = len(graphs)
    n_at = batch["pos"].shape[0] // n_g
    for g in sorted({0, 17 % n_g, n_g - 1}):
        single = {k: v.to(device) for k, v in collate([graphs[g]]).items()}
        e1, f1, _ = _energy_forces(model, single)
        assert _rel(e1, e0[g : g + 1]) < 1e-5 and _rel(f1, f0[n_at * g : n_at * (g + 1)]) < 1e-5
    b = dict(batch)
    b["pos"] = batch["pos"] + torch.tensor([0.37, -1.2, 2.9], device=device)
    e2, f2, _ = _energy_forces(model, b)
    assert _rel(e2, e0) < 1e-5 and _rel(f2, f0) < 1e-5
    assert f0.reshape(n_g, n_at, 3).sum(1).abs().max().item() < 1e-4 * f0.abs().max().item() * n_at  # forces sum to zero per crystal


def test_full_size_determinism(c3):
    """Same inputs, same bits: energies, forces and every parameter gradient (no atomics anywhere)."""
    model, batch, _ = c3
    a = _energy_forces(model, batch, grads=True)
    b = _energy_forces(model, batch, grads=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


def test_full_size_directional_derivative(c3, device):
    """-F . d matches the central difference of the (fp64-accumulated) energy along a direction that mixes a
    random displacement with the force field itself (so that the energy change stands clear of the fp32
    rounding of a total over thousands of atoms at every size)."""
    model, batch, _ = c3
    _, f0, _ = _energy_forces(model, batch)
    d = torch.randn(batch["pos"].shape, generator=torch.Generator().manual_seed(2)).to(device)
    d = d / d.norm() - f0 / f0.norm()
    d = d / d.norm()
    eps = 2e-2

    def total(shift):
        b = dict(batch)
        b["pos"] = batch["pos"] + shift * d
        with torch.no_grad():
            return model(b)["atomic_energy"].double().sum().item()  # per-atom energies, summed in fp64

    fd = (total(eps) - total(-eps)) / (2 * eps)
    an = -(f0.double() * d.double()).sum().item()
    assert abs(fd - an) < 2e-3 * max(abs(an), f0.double().norm().item() * 1e-1)


@pytest.mark.parametrize("cfg", [C3, C4], ids=["c3", "c4"])
def test_full_size_conv_is_bilinear(device, cfg):
    """The fused gather + uvu tensor product + scatter through the C ABI at the last layer's shapes:
    linear in the per-edge weights and in the node features, and zero edges-of-a-node give zero rows."""
    from matten_amd import keys as K
    from matten_amd import ops
    from matten_amd.data.crystal import synthetic_batch
    from matten_amd.nn.utils import UVUTensorProduct

    batch, _ = synthetic_batch(cfg["n"], cfg["a"], cfg["graphs"] if cfg is C3 else 8, seed0=1000, r_cut=5.0)
    ei = batch[K.EDGE_INDEX].to(device)
    n, e = batch[K.POSITIONS].shape[0], ei.shape[1]
    lmax = 3 if "3o" in cfg["sh"] else 2
    in1 = cfg["irreps"]
    tp = UVUTensorProduct(in1, cfg["sh"], in1, mlp_input_size=8).tp
    order = ops.EdgeOrdering(ei, n)
    g = torch.Generator().manual_seed(3)
    x1, x2 = (torch.randn(n, tp.irreps_in1.dim, generator=g).to(device) for _ in range(2))
    w1, w2 = (torch.randn(e, tp.weight_numel, generator=g).to(device) for _ in range(2))
    y = ops.spherical_harmonics(torch.randn(e, 3, generator=g).to(device), lmax)
    f = lambda x, w: tp(x, y, w, order, 0.19)  # noqa: E731
    ref = f(x1, w1)
    scale = ref.abs().max().item()
    assert (f(x1, 0.5 * w1 - 2.0 * w2) - (0.5 * ref - 2.0 * f(x1, w2))).abs().max().item() < 2e-5 * scale
    assert (f(x1 + 3.0 * x2, w1) - (ref + 3.0 * f(x2, w1))).abs().max().item() < 2e-5 * scale
    # drop every edge into node 0: its row becomes exactly zero, the others stay bit-identical
    keep = ei[1] != 0
    order2 = ops.EdgeOrdering(ei[:, keep].contiguous(), n)
    out2 = tp(x1, y[keep].contiguous(), w1[keep].contiguous(), order2, 0.19)
    assert out2[0].abs().max().item() == 0.0 and torch.equal(out2[1:], ref[1:])


def test_full_size_edge_ordering_structure(device):
    """tfc_csr_build at config 4's size: row_ptr is the exclusive scan of the key histogram, eid is a
    permutation that is ascending inside every row (hence the fixed summation order), nbr is the other
    endpoint; all bit-exact."""
    from matten_amd import keys as K
    from matten_amd import ops
    from matten_amd.data.crystal import synthetic_batch

    batch, _ = synthetic_batch(C4["n"], C4["a"], 32, seed0=1000, r_cut=5.0)
    ei = batch[K.EDGE_INDEX]
    n, e = batch[K.POSITIONS].shape[0], ei.shape[1]
    assert n == 6912 and e > 190000
    o = ops.EdgeOrdering(ei.to(device), n)
    o.validate()
    for key_row, (rp, eid, nbr) in ((1, o.by_dst), (0, o.by_src)):
        rp, eid, nbr = rp.cpu().long(), eid.cpu().long()[:e], nbr.cpu().long()[:e]
        counts = torch.bincount(ei[key_row], minlength=n)
        assert torch.equal(rp, torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))
        assert torch.equal(torch.sort(eid).values, torch.arange(e))
        assert torch.equal(ei[key_row][eid], torch.repeat_interleave(torch.arange(n), counts))
        assert torch.equal(nbr, ei[1 - key_row][eid])
        inner = torch.ones(e, dtype=torch.bool)
        inner[rp[1:-1][rp[1:-1] < e]] = False  # first edge of each row
        assert bool((eid[1:] > eid[:-1])[inner[1:]].all())
    # the neighbour list itself: every (i, j, S) has its mirror (j, i, -S); edge lengths below the cutoff
    sh = batch[K.EDGE_CELL_SHIFT].long()
    key = lambda i, j, s: ((i * n + j) * 27 + (s[:, 0] + 1) * 9 + (s[:, 1] + 1) * 3 + (s[:, 2] + 1))  # noqa: E731
    assert int(sh.abs().max()) <= 1
    assert torch.equal(torch.sort(key(ei[0], ei[1], sh)).values, torch.sort(key(ei[1], ei[0], -sh)).values)
    assert int(batch["num_neigh"].sum()) == e
