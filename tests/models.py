"""Shared by the model-level tests: the two target backbones (energy model of
scripts/configs/minimal.yaml, atomic tensor model of minimal_atomic.yaml) built from either the
oracle or matten_amd, on the reference's 20 matbench structures (tests/golden fixture)."""
from __future__ import annotations

import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "matbench_dielectric_n20_structures.json")

# model.backbone_hparams of scripts/configs/minimal.yaml:17-56
MINIMAL = dict(
    species_embedding_dim=32, irreps_edge_sh="0e + 1o + 2e", num_radial_basis=8, radial_basis_r_cut=5.0, num_layers=3,
    invariant_layers=2, invariant_neurons=64, average_num_neighbors="auto",
    conv_layer_irreps="32x0o + 32x0e + 16x1o + 16x1e + 8x2o + 8x2e", nonlinearity_type="gate", use_sc=True, batch_norm=False,
    resnet=True, conv_to_output_hidden_irreps_out="16x0e", reduce="sum",
)
# minimal_atomic.yaml:21-64
MINIMAL_ATOMIC = dict(MINIMAL, conv_to_output_hidden_irreps_out="16x0e + 8x1e + 4x2e", output_format="cartesian", output_formula="ij=ji")
MINIMAL_ATOMIC.pop("reduce")


def load_dataset():
    from matten_amd.data.structures import dataset_hparams, graphs_from_records

    records = json.load(open(FIXTURE))
    graphs = graphs_from_records(records, 5.0)
    targets = torch.tensor([r["n"] for r in records], dtype=torch.float64)
    return graphs, targets, dataset_hparams(graphs)


def make_batch(graphs, ids, device, dtype):
    from matten_amd.data.crystal import collate

    b = collate([graphs[i] for i in ids])
    return {k: (v.to(dtype) if v.is_floating_point() else v).to(device) for k, v in b.items()}


def build(impl, kind, dataset_hp):
    if impl == "oracle":
        from oracle import eigenn_ref as F
    else:
        from matten_amd import model_factory as F
    if kind == "energy":
        return F.create_energy_model(dict(MINIMAL), dataset_hp)
    return F.create_atomic_tensor_model(dict(MINIMAL_ATOMIC), dataset_hp)


def train_trajectory(model, graphs, targets, device, dtype, epochs=2, batch_size=10, lr=1e-3):
    """What BaseModel.shared_step + Lightning do for minimal.yaml: L1Loss between
    total_energy.reshape(-1) and y["n"] (nequip_energy_model.py:46-47, task.py:422-424), Adam(lr 1e-3),
    batch 10; the shuffle is replaced by a fixed order so that two implementations see the same batches."""
    model = model.to(device=device, dtype=dtype)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    order = list(range(len(graphs)))
    losses = []
    for _ in range(epochs):
        for s in range(0, len(order), batch_size):
            ids = order[s : s + batch_size]
            out = model(make_batch(graphs, ids, device, dtype))
            loss = torch.nn.functional.l1_loss(out["total_energy"].reshape(-1), targets[ids].to(device=device, dtype=dtype))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
    return losses


LSDI_FIXTURE = os.path.join(ROOT, "tests", "golden", "lsdi_nmr_n20_structures.json")


def load_lsdi():
    """configs[1]'s data: one data point per NMR tensor (eigenn/dataset/LSDI.py:69-88: node mask with the
    tensor's site set, tensor symmetrised), graphs at r_cut 5; returns graphs, per-point masks [N] bool and
    targets [1,3,3] fp64, and the dataset hparams (LSDI.py get_to_model_info = matbench's)."""
    import numpy as np

    from matten_amd.data.structures import dataset_hparams, graphs_from_records

    entries = json.load(open(LSDI_FIXTURE))
    per_entry = graphs_from_records(entries, 5.0)
    graphs, masks, targets = [], [], []
    for e, g in zip(entries, per_entry):
        for t, site in zip(e["tensors"], e["ind"]):
            m = np.zeros(len(e["pos"]), dtype=bool)
            m[site] = True
            t = torch.tensor(t, dtype=torch.float64)
            graphs.append(g)
            masks.append(torch.from_numpy(m))
            targets.append(((t + t.T) / 2).reshape(1, 3, 3))
    return graphs, masks, targets, dataset_hparams(graphs)


def tensor_train_trajectory(model, graphs, masks, targets, device, dtype, epochs=2, batch_size=10, lr=1e-3):
    """minimal_atomic.yaml's loop: L1Loss between tensor_output [n_selected,3,3] and the targets, Adam."""
    model = model.to(device=device, dtype=dtype)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for _ in range(epochs):
        for s in range(0, len(graphs), batch_size):
            ids = list(range(s, min(s + batch_size, len(graphs))))
            b = make_batch(graphs, ids, device, dtype)
            b["node_masks"] = torch.cat([masks[i] for i in ids]).to(device)
            y = torch.cat([targets[i] for i in ids]).to(device=device, dtype=dtype)
            out = model(b)["tensor_output"]
            loss = torch.nn.functional.l1_loss(out, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
    return losses
