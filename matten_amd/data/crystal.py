"""Synthetic periodic crystals and PyG-style batching for benchmarks and tests.

Generator: jittered simple-cubic supercell, n x n x n sites, lattice constant
a, cubic cell n*a, full PBC, positions = lattice + N(0, sigma^2) per coordinate from
`numpy.random.default_rng(seed)`, species uniform from {1, 6, 8, 14}; re-drawn if any pair
distance falls within 1e-6 of r_cut so that neighbour-list parity stays away from rounding.

Batching follows torch_geometric's collate as used by the reference's DataLoader
(eigenn/data/datamodule.py): node tensors concatenated, `edge_index` offset by the running
node count, `batch` = graph id per node, per-graph `cell` stacked to [3G, 3]
(eigenn/data/data.py:91, 158-171).  Tensors are fp32 / int64 (eigenn/data/_dtype.py:3-4).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .. import keys as K
from .neighbor import neighbor_list

SPECIES = (1, 6, 8, 14)


def make_crystal(n: int, a: float, sigma: float = 0.05, r_cut: float = 5.0, seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    grid = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64) * a
    cell = np.eye(3) * (n * a)
    for _ in range(100):
        pos = grid + rng.normal(0.0, sigma, size=grid.shape)
        z = rng.choice(SPECIES, size=len(grid))
        ei, shifts, num_neigh = neighbor_list(pos, cell, True, r_cut)
        vec = pos[ei[1]] - pos[ei[0]] + shifts.astype(np.float64) @ cell
        d = np.linalg.norm(vec, axis=1)
        # also make sure nothing sits just outside the cutoff
        ei2, sh2, _ = neighbor_list(pos, cell, True, r_cut + 2e-6)
        if len(d) == ei2.shape[1] and np.all(np.abs(d - r_cut) > 1e-6):
            return dict(pos=pos, cell=cell, atomic_numbers=z, edge_index=ei, edge_cell_shift=shifts, num_neigh=num_neigh)
    raise RuntimeError("could not draw a crystal away from the cutoff")


def collate(graphs: Sequence[Dict[str, np.ndarray]], device="cpu") -> Dict[str, torch.Tensor]:
    pos, cell, z, ei, sh, batch, nn = [], [], [], [], [], [], []
    off = 0
    ptr = [0]
    for g, gr in enumerate(graphs):
        n = len(gr["pos"])
        pos.append(torch.as_tensor(gr["pos"], dtype=torch.float32))
        cell.append(torch.as_tensor(gr["cell"], dtype=torch.float32).reshape(3, 3))
        z.append(torch.as_tensor(gr["atomic_numbers"], dtype=torch.int64))
        ei.append(torch.as_tensor(gr["edge_index"], dtype=torch.int64) + off)
        sh.append(torch.as_tensor(gr["edge_cell_shift"], dtype=torch.float32))
        nn.append(torch.as_tensor(gr["num_neigh"], dtype=torch.float32))
        batch.append(torch.full((n,), g, dtype=torch.int64))
        off += n
        ptr.append(off)
    out = {
        K.POSITIONS: torch.cat(pos),
        K.CELL: torch.cat(cell),  # [3G, 3]
        K.ATOMIC_NUMBERS: torch.cat(z),
        K.EDGE_INDEX: torch.cat(ei, dim=1),
        K.EDGE_CELL_SHIFT: torch.cat(sh),
        K.BATCH: torch.cat(batch),
        "num_neigh": torch.cat(nn),
        "ptr": torch.tensor(ptr, dtype=torch.int64),  # PyG's graph boundaries: nodes of graph g are ptr[g]:ptr[g+1]
    }
    return {k: v.to(device) for k, v in out.items()}


def synthetic_batch(n: int, a: float, batch: int, seed0: int = 1000, r_cut: float = 5.0, sigma: float = 0.05, device="cpu"):
    graphs = [make_crystal(n, a, sigma, r_cut, seed0 + g) for g in range(batch)]
    return collate(graphs, device), graphs


def collate_gpu(graphs: Sequence[Dict[str, np.ndarray]], r_cut: float, device) -> Dict[str, torch.Tensor]:
    """The same batch as `collate`, with the neighbour lists of all structures built on the device
    in one go (`neighbor_gpu.neighbor_list_batch`, row f2) instead of being read from the graphs:
    `graphs` need only `pos`, `cell`, `atomic_numbers` (and optionally `pbc`).  The edge groupings the
    convolution layers need are derived from the list's symmetry and registered for the returned
    `edge_index` tensor, so the layers do not sort it again.  Tensors equal `collate`'s bit for bit."""
    from .neighbor_gpu import neighbor_list_batch

    device = torch.device(device)
    nl = neighbor_list_batch(graphs, r_cut, device)
    sizes = [len(g["pos"]) for g in graphs]
    out = {
        K.POSITIONS: torch.as_tensor(np.concatenate([np.asarray(g["pos"], dtype=np.float64) for g in graphs]), dtype=torch.float32),
        K.CELL: torch.as_tensor(np.concatenate([np.asarray(g["cell"], dtype=np.float64).reshape(3, 3) for g in graphs]), dtype=torch.float32),
        K.ATOMIC_NUMBERS: torch.as_tensor(np.concatenate([np.asarray(g["atomic_numbers"]) for g in graphs]), dtype=torch.int64),
        K.BATCH: torch.repeat_interleave(torch.arange(len(graphs), dtype=torch.int64), torch.tensor(sizes, dtype=torch.int64)),
    }
    out = {k: v.to(device) for k, v in out.items()}
    out[K.EDGE_INDEX] = nl["edge_index"]
    out[K.EDGE_CELL_SHIFT] = nl["edge_cell_shift"]
    out["num_neigh"] = nl["num_neigh"].to(torch.float32)
    out["ptr"] = nl["ptr"]
    out["edge_ordering"] = nl["ordering"]  # keeps the registered ordering alive with the batch
    return out


def pad_null_edges(batch: Dict[str, torch.Tensor], n_edges: int, k: int = 1000) -> Dict[str, torch.Tensor]:
    """The batch with its edge list padded to `n_edges` entries by NULL edges: node j -> node j across `k` cells, i.e.
    far beyond any cutoff, j running round the nodes (all on one node would make that node's segment the longest of the
    batch by far and the convolution kernels wait for the workgroup that owns it: +20 % on the backward, measured).
    Their radial embedding is exactly zero (the cutoff polynomial is multiplied by r < r_max,
    eigenn/nn/_nequip.py:69-78), hence - the radial network has no bias and ssp(0) = 0 - so are their tensor-product
    weights, messages and every gradient through them: energies, forces and parameter gradients are those of the
    unpadded batch, while all batches of a padded stream share one shape (matten_amd.graphs.GraphedStep)."""
    ei, sh = batch[K.EDGE_INDEX], batch[K.EDGE_CELL_SHIFT]
    e = int(ei.shape[1])
    if n_edges < e:
        raise ValueError(f"pad_null_edges: the batch has {e} edges, more than {n_edges}")
    out = dict(batch)
    pad = n_edges - e
    if pad:
        n_nodes = int(batch[K.POSITIONS].shape[0])
        own = (torch.arange(pad, dtype=ei.dtype, device=ei.device) % max(n_nodes, 1)).expand(2, pad)
        out[K.EDGE_INDEX] = torch.cat([ei, own], dim=1)
        far = torch.zeros(pad, 3, dtype=sh.dtype, device=sh.device)
        far[:, 0] = k
        out[K.EDGE_CELL_SHIFT] = torch.cat([sh, far], dim=0)
    return out
