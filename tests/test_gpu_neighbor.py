"""-m gpu: the neighbour list built on the device (tfc_nl_count / tfc_nl_fill / tfc_nl_ordering, row f2
of SURVEY.md section 8) against the oracle's brute-force enumeration, the host implementation and
the committed digests of the reference's own 20 matbench structures - bit-exact arrays (canonical
order), as the north star asks for edge-index construction.  Edge cases of
eigenn/data/data.py:400-528: self images kept, (i == j, S == 0) dropped, strict cutoff, ValueError
when a structure keeps no edge, non-periodic axes, atoms outside the cell, ragged batches."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def canon(ei, sh):
    return np.concatenate([np.asarray(ei).T, np.asarray(sh)], axis=1).astype(np.int64)


def host_batch(structs, r_max):
    """The host lists of the structures, concatenated the way PyG's collate offsets them."""
    from matten_amd.data.neighbor import neighbor_list

    ei, sh, nn, off = [], [], [], 0
    for s in structs:
        e, h, n = neighbor_list(s["pos"], s["cell"], s.get("pbc", True), r_max)
        ei.append(e + off)
        sh.append(h)
        nn.append(n)
        off += len(s["pos"])
    return np.concatenate(ei, 1), np.concatenate(sh), np.concatenate(nn)


def check_ordering(out, n_nodes):
    """tfc_nl_ordering must equal the sort-based tfc_csr_build on the same edge list."""
    from matten_amd import ops

    a = out["ordering"]
    b = ops.EdgeOrdering(out["edge_index"].clone(), n_nodes)
    for x, y in zip(a.by_dst + a.by_src, b.by_dst + b.by_src):
        assert torch.equal(x, y)


def test_matbench_structures_bit_exact(device):
    from matten_amd.data.neighbor_gpu import neighbor_list_batch

    data = json.load(open(os.path.join(HERE, "golden", "matbench_dielectric_n20_structures.json")))
    structs = [dict(pos=np.array(s["pos"]), cell=np.array(s["cell"])) for s in data]
    r_cut = data[0]["r_cut"]
    assert all(s["r_cut"] == r_cut for s in data)
    out = neighbor_list_batch(structs, r_cut, device)
    ei, sh = out["edge_index"].cpu().numpy(), out["edge_cell_shift"].cpu().numpy()
    ptr = out["ptr"].cpu().numpy()
    rp = out["row_ptr"].cpu().numpy()
    for g, s in enumerate(data):  # per structure: the committed digest of the brute-force set
        lo, hi = rp[ptr[g]], rp[ptr[g + 1]]
        assert hi - lo == s["n_edges"]
        c = canon(ei[:, lo:hi] - ptr[g], sh[lo:hi])
        assert hashlib.sha256(np.ascontiguousarray(c).tobytes()).hexdigest() == s["edges_sha256"]
        assert out["num_neigh"][ptr[g] : ptr[g + 1]].cpu().tolist() == s["num_neigh"]
    assert np.all(np.diff(ei[0]) >= 0)
    check_ordering(out, int(ptr[-1]))


@pytest.mark.parametrize("pbc", [True, (True, False, True), False])
def test_random_triclinic_batch_against_bruteforce(device, pbc):
    from matten_amd.data.neighbor_gpu import neighbor_list_batch
    from oracle.eigenn_ref import neighbor_list_bruteforce

    structs = []
    for seed in range(6):
        g = np.random.default_rng(seed)
        cell = np.eye(3) * 4.0 + g.normal(0, 0.8, (3, 3))
        pos = g.uniform(-2, 6, (5 + 3 * seed, 3))  # ragged, not wrapped into the cell on purpose
        structs.append(dict(pos=pos, cell=cell, pbc=pbc))
    out = neighbor_list_batch(structs, 3.7, device)
    ei, sh = out["edge_index"].cpu().numpy(), out["edge_cell_shift"].cpu().numpy()
    ref_e, ref_s, off = [], [], 0
    for s in structs:
        e, h = neighbor_list_bruteforce(s["pos"], s["cell"], pbc if not isinstance(pbc, bool) else (pbc,) * 3, 3.7)
        ref_e.append(np.asarray(e) + off)
        ref_s.append(np.asarray(h))
        off += len(s["pos"])
    assert np.array_equal(canon(ei, sh), canon(np.concatenate(ref_e, 1), np.concatenate(ref_s)))
    assert out["edge_index"].dtype == torch.int64 and out["edge_cell_shift"].dtype == torch.float32
    check_ordering(out, off)
    # the image pruning (graph_box) only removes candidates the distance test would reject
    full = neighbor_list_batch(structs, 3.7, device, with_ordering=False, prune_images=False)
    assert torch.equal(full["edge_index"], out["edge_index"]) and torch.equal(full["edge_cell_shift"], out["edge_cell_shift"])


def test_self_images_strict_cutoff_and_empty(device):
    from matten_amd.data.neighbor_gpu import neighbor_list_batch

    cell = np.eye(3) * 3.0
    out = neighbor_list_batch([dict(pos=np.zeros((1, 3)), cell=cell)], 3.0 + 1e-9, device)
    ei, sh = out["edge_index"].cpu().numpy(), out["edge_cell_shift"].cpu().numpy()
    assert ei.shape[1] == 6 and np.all(ei == 0) and not np.any(np.all(sh == 0, axis=1))
    check_ordering(out, 1)
    with pytest.raises(ValueError, match="no edges remain"):
        neighbor_list_batch([dict(pos=np.zeros((1, 3)), cell=cell)], 3.0, device)  # distance == cutoff is excluded
    with pytest.raises(ValueError, match="no edges remain"):  # one empty structure in a batch is an error, as per structure in the reference
        neighbor_list_batch(
            [dict(pos=np.zeros((1, 3)), cell=cell), dict(pos=np.array([[0.0, 0, 0], [9.0, 0, 0]]), cell=cell, pbc=False)], 3.5, device
        )
    with pytest.raises(RuntimeError, match="ROCm device"):
        neighbor_list_batch([dict(pos=np.zeros((1, 3)), cell=cell)], 3.5, "cpu")


@pytest.mark.parametrize("n,a,batch", [(4, 2.5, 8), (6, 2.5, 2), (12, 2.1, 1)])
def test_synthetic_crystals_equal_host_list(device, n, a, batch):
    """The bench configurations (config 3 / 4 / 5 cells): device list == host list, array for array,
    and the orderings derived from symmetry == the sorted ones."""
    from matten_amd.data.neighbor_gpu import neighbor_list_batch

    structs = []
    for b in range(batch):
        g = np.random.default_rng(1000 + b)
        grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * a
        structs.append(dict(pos=grid + g.normal(0.0, 0.05, size=grid.shape), cell=np.eye(3) * (n * a)))
    out = neighbor_list_batch(structs, 5.0, device)
    ei, sh, nn = host_batch(structs, 5.0)
    assert np.array_equal(out["edge_index"].cpu().numpy(), ei)
    assert np.array_equal(out["edge_cell_shift"].cpu().numpy(), sh.astype(np.float32))
    assert np.array_equal(out["num_neigh"].cpu().numpy(), nn)
    check_ordering(out, sum(len(s["pos"]) for s in structs))
    # size-independent properties: symmetric list, every edge inside the cutoff
    pos = np.concatenate([s["pos"] for s in structs])
    cellv = n * a
    d = np.linalg.norm(pos[ei[1]] - pos[ei[0]] + sh * cellv, axis=1)
    assert d.max() < 5.0 and d.min() > 0.0
    fwd = {tuple(r) for r in canon(ei, sh)[:: max(1, ei.shape[1] // 5000)]}
    rev = {(j, i, -x, -y, -z) for i, j, x, y, z in fwd}
    allset = set(map(tuple, canon(ei, sh)))
    assert rev <= allset


def test_ordering_feeds_the_layer_cache(device):
    """The ordering handed over by the neighbour list is the one ops.edge_ordering returns for that tensor."""
    from matten_amd import ops
    from matten_amd.data.neighbor_gpu import neighbor_list_batch

    g = np.random.default_rng(3)
    st = dict(pos=g.uniform(0, 6, (20, 3)), cell=np.eye(3) * 6.0)
    out = neighbor_list_batch([st], 4.0, device)
    assert ops.edge_ordering(out["edge_index"], 20) is out["ordering"]


def test_ordering_flags_asymmetric_list(device):
    """A list that is not symmetric is reported (status 1) and the general builder is used instead."""
    from matten_amd import ops

    ei = torch.tensor([[0, 0, 1, 2], [1, 2, 0, 1]], dtype=torch.int64, device=device)  # 2->1 has no 1->2
    rp = torch.tensor([0, 2, 3, 4], dtype=torch.int32, device=device)
    o = ops.EdgeOrdering.from_neighbor_list(ei, 3, rp)
    ref = ops.EdgeOrdering(ei.clone(), 3)
    for x, y in zip(o.by_dst + o.by_src, ref.by_dst + ref.by_src):
        assert torch.equal(x, y)


def test_collate_gpu_equals_host_collate_and_layers_reuse_the_ordering(device):
    """A batch assembled with the device neighbour list is tensor-for-tensor the host-collated batch,
    and a conv stack run on either gives the same energies and forces bit for bit (same edge order,
    same groupings: the summation orders are identical)."""
    from matten_amd import keys as K, ops
    from matten_amd.data.crystal import collate, collate_gpu, make_crystal

    graphs = [make_crystal(3, 2.5, seed=40 + g) for g in range(3)]
    a = collate(graphs, device)
    b = collate_gpu([{k: g[k] for k in ("pos", "cell", "atomic_numbers")} for g in graphs], 5.0, device)
    for k in (K.POSITIONS, K.CELL, K.ATOMIC_NUMBERS, K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.BATCH, "num_neigh", "ptr"):
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    assert ops.edge_ordering(b[K.EDGE_INDEX], a[K.POSITIONS].shape[0]) is b["edge_ordering"]

    import matten_amd.nn as H
    from matten_amd.harness import ConvStack, species_index

    torch.manual_seed(35)
    model = ConvStack(H, 16, 2, "8x0e+8x1o+8x2e", 2, 16, 2, 5.0, 28.0).to(device)
    outs = []
    for batch in (a, b):
        bb = {k: v for k, v in batch.items() if k != "edge_ordering"}
        bb["species_index"] = species_index(bb[K.ATOMIC_NUMBERS])
        pos = bb[K.POSITIONS].clone().requires_grad_(True)
        bb[K.POSITIONS] = pos
        _, energy, _ = model(bb, len(graphs))
        (f,) = torch.autograd.grad(energy.sum(), pos)
        outs.append((energy.detach().clone(), f.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
