"""Real structures -> graphs -> batches, without pymatgen / PyG.

`graphs_from_records` turns records {cell [3,3], pos [N,3] Cartesian, atomic_numbers [N]} (the
committed fixture tests/golden/matbench_dielectric_n20_structures.json holds the reference's 20
matbench structures in this form) into the per-structure arrays the reference's `Crystal.from_points`
produces (eigenn/data/data.py:330-398: full PBC, neighbour list at r_cut, num_neigh) and
`dataset_hparams` computes what `get_to_model_info` hands to the model factory
(eigenn/dataset/matbench.py:148-162): the set of species and the mean neighbour count per atom.
Batching is `crystal.collate` (PyG-style)."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .neighbor import neighbor_list


def graphs_from_records(records: Sequence[dict], r_cut: float) -> List[Dict[str, np.ndarray]]:
    graphs = []
    for rec in records:
        pos = np.asarray(rec["pos"], dtype=np.float64)
        cell = np.asarray(rec["cell"], dtype=np.float64).reshape(3, 3)
        ei, shifts, num_neigh = neighbor_list(pos, cell, True, r_cut)
        graphs.append(dict(pos=pos, cell=cell, atomic_numbers=np.asarray(rec["atomic_numbers"], dtype=np.int64),
                           edge_index=ei, edge_cell_shift=shifts, num_neigh=num_neigh))
    return graphs


def dataset_hparams(graphs: Sequence[Dict[str, np.ndarray]]) -> dict:
    species = sorted({int(z) for g in graphs for z in g["atomic_numbers"]})
    num_neigh = np.concatenate([np.asarray(g["num_neigh"], dtype=np.float32) for g in graphs])
    return {"allowed_species": tuple(species), "average_num_neighbors": float(num_neigh.mean(dtype=np.float32))}
