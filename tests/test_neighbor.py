"""CPU: neighbour lists.  Bit-exact (as a canonical set) against the oracle's brute-force
enumeration and against the committed fixture built from the reference's own 20 matbench
structures (tests/golden/make_structures_fixture.py); edge cases of
eigenn/data/data.py:400-528 (self images kept, (i==j, S==0) dropped, strict cutoff, ValueError
when nothing remains, non-periodic axes)."""
import hashlib
import json
import os

import numpy as np
import pytest

from matten_amd.data.neighbor import neighbor_list
from oracle.eigenn_ref import neighbor_list_bruteforce

HERE = os.path.dirname(os.path.abspath(__file__))


def canon(ei, sh):
    return np.concatenate([ei.T, sh], axis=1).astype(np.int64)


def test_matbench_fixture_bit_exact():
    data = json.load(open(os.path.join(HERE, "golden", "matbench_dielectric_n20_structures.json")))
    assert [len(s["pos"]) for s in data] == [9, 36, 24, 24, 6, 24, 24, 20, 6, 9, 14, 28, 10, 5, 16, 8, 12, 20, 12, 24]
    for s in data:
        ei, sh, nn = neighbor_list(np.array(s["pos"]), np.array(s["cell"]), True, s["r_cut"])
        assert ei.shape[1] == s["n_edges"]
        assert nn.tolist() == s["num_neigh"]
        assert hashlib.sha256(np.ascontiguousarray(canon(ei, sh)).tobytes()).hexdigest() == s["edges_sha256"]
        assert np.all(np.diff(ei[0]) >= 0)  # non-decreasing first index, as ASE delivers


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("pbc", [True, (True, False, True), False])
def test_random_triclinic_against_bruteforce(seed, pbc):
    g = np.random.default_rng(seed)
    cell = np.eye(3) * 4.0 + g.normal(0, 0.8, (3, 3))
    pos = g.uniform(-2, 6, (11, 3))  # not wrapped into the cell on purpose
    a = neighbor_list(pos, cell, pbc, 3.7)
    b = neighbor_list_bruteforce(pos, cell, pbc if not isinstance(pbc, bool) else (pbc,) * 3, 3.7)
    assert np.array_equal(canon(a[0], a[1]), canon(b[0], b[1]))


def test_self_images_and_strict_cutoff():
    cell = np.eye(3) * 3.0
    ei, sh, nn = neighbor_list(np.zeros((1, 3)), cell, True, 3.0 + 1e-9)
    assert ei.shape[1] == 6 and np.all(ei == 0) and not np.any(np.all(sh == 0, axis=1))
    with pytest.raises(ValueError, match="no edges remain"):
        neighbor_list(np.zeros((1, 3)), cell, True, 3.0)  # distance == cutoff is excluded
    with pytest.raises(ValueError, match="no edges remain"):
        neighbor_list(np.array([[0.0, 0, 0], [9.0, 0, 0]]), cell, False, 3.0)


def test_synthetic_crystal_generator_is_reproducible():
    from matten_amd.data.crystal import make_crystal

    a, b = make_crystal(3, 2.5, seed=5), make_crystal(3, 2.5, seed=5)
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["edge_index"], b["edge_index"])
    assert len(a["pos"]) == 27 and set(a["atomic_numbers"]) <= {1, 6, 8, 14}
    d = np.linalg.norm(a["pos"][a["edge_index"][1]] - a["pos"][a["edge_index"][0]] + a["edge_cell_shift"] @ a["cell"], axis=1)
    assert d.max() < 5.0 - 1e-6 and d.min() > 0.5
