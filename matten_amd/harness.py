"""Minimal end-to-end harness around the hot path, shared by the parity tests and bench.py:
species lookup -> SphericalHarmonicEdgeAttrs -> RadialBasisEdgeEncoding -> L x
PointConvMessagePassing -> fixed random linear read-out to a per-graph "energy".

It reproduces what the reference's model factory wires around the layer
(eigenn/model_factory/nequip_energy_model.py:63-176: same layer order and kwargs) without
Lightning; the embedding and read-out are plain torch on both sides because they are outside
the path.  `modules` is the namespace providing the layer classes
(matten_amd.nn for the HIP path; the tests pass the oracle's namespace instead).
"""
from __future__ import annotations

import torch

from . import keys as K
from .irreps import Irreps

SPECIES_TABLE = {1: 0, 6: 1, 8: 2, 14: 3}


def species_index(z: torch.Tensor) -> torch.Tensor:
    lut = torch.full((max(SPECIES_TABLE) + 1,), -1, dtype=torch.long)
    for k, v in SPECIES_TABLE.items():
        lut[k] = v
    return lut[z.cpu()].to(z.device)


class ConvStack(torch.nn.Module):
    def __init__(self, modules, emb: int, lmax: int, conv_irreps: str, layers: int, hidden: int = 64, n_hidden: int = 2,
                 r_cut: float = 5.0, avg_num_neighbors: float = 28.0, num_basis: int = 8, activation_type: str = "gate"):
        super().__init__()
        M = modules
        self.embedding = torch.nn.Embedding(len(SPECIES_TABLE), emb)
        irreps = {K.NODE_FEATURES: f"{emb}x0e", K.NODE_ATTRS: f"{emb}x0e", K.EDGE_INDEX: None, K.POSITIONS: "1o"}
        self.spharm_edges = M.SphericalHarmonicEdgeAttrs(irreps_edge_sh=lmax, irreps_in=irreps)
        self.radial_basis = M.RadialBasisEdgeEncoding(
            basis_kwargs={"num_basis": num_basis, "r_max": r_cut}, cutoff_kwargs={"r_max": r_cut}, irreps_in=self.spharm_edges.irreps_out
        )
        cur = self.radial_basis.irreps_out
        self.convs = torch.nn.ModuleList()
        for _ in range(layers):
            layer = M.PointConvMessagePassing(
                irreps_in=cur,
                conv_layer_irreps=conv_irreps,
                message_kwargs={"fc_num_hidden_layers": n_hidden, "fc_hidden_size": hidden},
                update_kwargs={"use_self_connection": True, "avg_num_neighbors": avg_num_neighbors},
                activation_type=activation_type,
                use_resnet=True,
            )
            self.convs.append(layer)
            cur = layer.irreps_out
        self.d_out = Irreps(str(cur[K.NODE_FEATURES])).dim
        g = torch.Generator().manual_seed(7)
        self.register_buffer("readout", torch.randn(self.d_out, generator=g) / self.d_out**0.5)
        self.classes_hint = True  # False: the layers find the row classes of the attributes themselves

    def forward(self, batch, n_graphs: int):
        data = dict(batch)
        # embedding lookup as one-hot @ table: same values, and a backward that is one small GEMM
        # instead of torch's single-block embedding_backward kernel
        w = self.embedding.weight
        x = torch.nn.functional.one_hot(data["species_index"], w.shape[0]).to(w.dtype) @ w
        data[K.NODE_FEATURES] = x
        data[K.NODE_ATTRS] = x
        # the attributes are an embedding of the species: tell the layers their classes (an
        # extension the oracle's layers ignore)
        if self.classes_hint:
            data["species_index"].tfc_num_classes = int(w.shape[0])
            data[K.NODE_ATTRS_CLASSES] = data["species_index"]
        data = self.spharm_edges(data)
        data = self.radial_basis(data)
        for layer in self.convs:
            data = layer(data)
        feats = data[K.NODE_FEATURES]
        per_atom = feats @ self.readout.to(feats.dtype)
        energy = torch.zeros(n_graphs, dtype=feats.dtype, device=feats.device).index_add_(0, data[K.BATCH], per_atom)
        return feats, energy, data
