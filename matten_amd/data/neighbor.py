"""Periodic neighbour list with the semantics of the reference's
`neighbor_list_and_relative_vec` (eigenn/data/data.py:400-528), which wraps
`ase.neighborlist.primitive_neighbor_list("ijS", pbc, cell, pos, cutoff=r_max,
self_interaction=True, use_scaled_positions=False)` and then drops the (i == j, S == 0) edge.

Result = the set of all (i, j, S) with |pos[j] - pos[i] + S @ cell| < r_max (strict, fp64),
returned in canonical order (i, j, Sx, Sy, Sz).  ASE's own order inside equal `i` comes from
an unstable argsort and is not reproducible, so parity with the reference is equality as a
set; `edge_index[0]` is non-decreasing here as it is there.
Host side (numpy, fp64): this is dataset preprocessing in the reference too.
"""
from __future__ import annotations

import itertools
import math
from typing import Sequence, Tuple, Union

import numpy as np


def image_table(pos: np.ndarray, cell: np.ndarray, pbc: Sequence[bool], r_max: float) -> np.ndarray:
    """The periodic images [S,3] (int64, ascending lexicographic, (0,0,0) included) that can hold a
    neighbour within r_max of an atom of this structure."""
    n = len(pos)
    # images needed along axis a: the cutoff sphere spans ceil(r_max / h_a) cells, h_a = distance
    # between the cell faces normal to reciprocal vector a; +1 because atoms need not be wrapped
    reps = [0, 0, 0]
    if any(pbc):
        vol = abs(np.linalg.det(cell))
        if vol < 1e-12:
            raise ValueError("periodic cell has zero volume")
        frac = pos @ np.linalg.inv(cell)
        spread = np.ceil(frac.max(0) - frac.min(0)).astype(int) if n else np.zeros(3, int)
        for a in range(3):
            if pbc[a]:
                h = vol / np.linalg.norm(np.cross(cell[(a + 1) % 3], cell[(a + 2) % 3]))
                reps[a] = int(math.ceil(r_max / h)) + int(spread[a])
    return np.array(list(itertools.product(*[range(-r, r + 1) for r in reps])), dtype=np.int64)


def neighbor_list(
    pos: np.ndarray, cell: np.ndarray, pbc: Union[bool, Sequence[bool]], r_max: float, chunk: int = 2048
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns (edge_index [2,E] int64, shifts [E,3] int64, num_neigh [N] int64)."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    if isinstance(pbc, (bool, np.bool_)):
        pbc = (bool(pbc),) * 3
    pbc = tuple(bool(b) for b in pbc)
    n = len(pos)
    r_max = float(r_max)

    shifts = image_table(pos, cell, pbc, r_max)
    shift_vec = shifts.astype(np.float64) @ cell  # [S,3]

    out_i, out_j, out_s = [], [], []
    r2 = r_max * r_max
    for s_idx in range(len(shifts)):
        sv = shift_vec[s_idx]
        zero = not shifts[s_idx].any()
        for a in range(0, n, chunk):
            d = pos[None, :, :] + sv[None, None, :] - pos[a : a + chunk, None, :]
            dist2 = np.einsum("ijk,ijk->ij", d, d)
            # strict `<` on the distance itself, as ASE compares distances, not squares
            ii, jj = np.nonzero(dist2 < r2 * (1 + 1e-12))
            if len(ii):
                keep = np.sqrt(dist2[ii, jj]) < r_max
                ii, jj = ii[keep] + a, jj[keep]
                if zero:
                    keep = ii != jj
                    ii, jj = ii[keep], jj[keep]
                out_i.append(ii)
                out_j.append(jj)
                out_s.append(np.broadcast_to(shifts[s_idx], (len(ii), 3)))
    if not out_i or sum(len(x) for x in out_i) == 0:
        raise ValueError("After eliminating self edges, no edges remain in this system.")
    i = np.concatenate(out_i)
    j = np.concatenate(out_j)
    s = np.concatenate(out_s)
    order = np.lexsort((s[:, 2], s[:, 1], s[:, 0], j, i))
    i, j, s = i[order], j[order], s[order]
    edge_index = np.vstack([i, j]).astype(np.int64)
    num_neigh = np.bincount(i, minlength=n).astype(np.int64)
    return edge_index, s.astype(np.int64), num_neigh
