"""Irreps -> Cartesian read-out (eigenn/nn/readout.py:12-60).

`CartesianTensor` restates the part of e3nn.io.CartesianTensor the reference uses: the irreps of
a rank-2 Cartesian formula and the change of basis Q [dim, 3, 3] with orthonormal rows,
to_cartesian(y) = sum_q y_q Q_q, from_cartesian(t) = <Q_q, t>.  Block l of Q is
sqrt(2l+1) * C^{1,1,l} (the real Clebsch-Gordan tensor of matten_amd.so3, vector index order
x, y, z).  e3nn derives its basis with ReducedTensorProducts; its per-irrep sign cannot be checked
without e3nn (parity unpinned) - the reference's own test of this module
(tests/nn/test_readout.py:7-22) is the sign-blind round trip, which tests/ repeats.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import keys as DataKey
from .. import so3
from ..irreps import Irreps
from ..module_irreps import ModuleIrreps

_FORMULA_LS = {"ij=ji": (0, 2), "ij=-ji": (1,), "ij": (0, 1, 2), "ij=ij": (0, 1, 2)}


class CartesianTensor(Irreps):
    """Irreps of a rank-2 Cartesian tensor formula, carrying the change of basis."""

    def __new__(cls, formula: str):
        key = formula.replace(" ", "")
        if key not in _FORMULA_LS:
            raise NotImplementedError(f"CartesianTensor: formula `{formula}` (rank-2 formulas {sorted(_FORMULA_LS)} are built)")
        ls = _FORMULA_LS[key]
        self = super().__new__(cls, "+".join(f"1x{l}e" for l in ls))
        self.formula = formula
        q = np.concatenate([np.transpose((2 * l + 1) ** 0.5 * so3.real_cg(1, 1, l), (2, 0, 1)) for l in ls], axis=0)
        self.change_of_basis = torch.from_numpy(q)  # fp64 [dim, 3, 3]
        return self

    def _q(self, like: torch.Tensor) -> torch.Tensor:
        return self.change_of_basis.to(device=like.device, dtype=like.dtype)

    def from_cartesian(self, t: torch.Tensor) -> torch.Tensor:
        return torch.einsum("qij,...ij->...q", self._q(t), t)

    def to_cartesian(self, y: torch.Tensor) -> torch.Tensor:
        return torch.einsum("qij,...q->...ij", self._q(y), y)


class IrrepsToCartesianTensor(ModuleIrreps, torch.nn.Module):
    def __init__(self, irreps_in: Dict[str, Irreps], formula: str = "ij=ji", field: str = DataKey.NODE_FEATURES,
                 out_field: Optional[str] = None):
        super().__init__()
        self.formula = formula
        self.field = field
        self.out_field = field if out_field is None else out_field
        # the output is no longer an irreps tensor, so nothing is added to irreps_out
        self.init_irreps(irreps_in, required_keys_irreps_in=[field])
        self.ct = CartesianTensor(formula)
        assert self.irreps_in[field] == self.ct, (
            f"input irreps of {field} is {self.irreps_in[field]}, not equal to the irreps of the target irreps {self.ct}"
        )
        self.register_buffer("_q", self.ct.change_of_basis.to(torch.float32), persistent=False)

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        v = data[self.field]
        # 9 outputs per selected atom from 6 (or 9) inputs: a [n, dim] x [dim, 9] product, left to torch
        data[self.out_field] = (v @ self._q.to(v.dtype).reshape(self._q.shape[0], 9)).reshape(*v.shape[:-1], 3, 3)
        return data

    def extra_repr(self):
        return f"{self.field} -> {self.out_field}, formula: {self.formula}"
