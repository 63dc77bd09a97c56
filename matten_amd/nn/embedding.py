"""Species embedding (eigenn/nn/embedding.py:12-75, 172-225).

NODE_ATTRS and NODE_FEATURES both receive the learnable embedding of the species index, as in
the reference.  Two things differ in mechanism, not in values: the lookup is a one-hot product
(its backward is one small dense product instead of torch's serial embedding_backward), and the
species index is also published under keys.NODE_ATTRS_CLASSES so that every conv layer's self
connection can contract over the few distinct attribute rows without inspecting the tensor.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .. import keys as DataKey
from ..irreps import Irreps
from ..module_irreps import ModuleIrreps


class _AtomicNumberToIndex(torch.nn.Module):
    """Sorted allowed atomic numbers -> 0..n-1; anything else raises RuntimeError."""

    def __init__(self, allowed_atomic_numbers: List[int]):
        super().__init__()
        allowed = torch.as_tensor(sorted(int(z) for z in allowed_atomic_numbers), dtype=torch.long)
        self.num_species = int(allowed.numel())
        self.min_z, self.max_z = int(allowed[0]), int(allowed[-1])
        table = torch.full((self.max_z - self.min_z + 1,), -1, dtype=torch.long)
        table[allowed - self.min_z] = torch.arange(self.num_species)
        # buffer names of the reference's module (embedding.py:186-194), so its checkpoints load strictly
        self.register_buffer("_min_Z", allowed.min())
        self.register_buffer("_max_Z", allowed.max())
        self.register_buffer("_num_species", torch.as_tensor(self.num_species))
        self.register_buffer("_Z_to_index", table)

    def forward(self, atomic_numbers: torch.Tensor) -> torch.Tensor:
        lo, hi = int(atomic_numbers.min()), int(atomic_numbers.max())  # host sync, once per batch
        if lo < self.min_z or hi > self.max_z:
            raise RuntimeError(
                f"Invalid atomic numbers. Expect atomic numbers to be in the range [{self.min_z}, {self.max_z}], "
                f"but got min {lo} and max {hi}"
            )
        index = self._Z_to_index[atomic_numbers - self.min_z]
        if int(index.min()) < 0:
            bad = atomic_numbers[index < 0][0].item()
            raise RuntimeError(f"Expect atomic numbers to be one of the allowed species, got invalid atomic number `{bad}`.")
        return index


class SpeciesEmbedding(ModuleIrreps, torch.nn.Module):
    def __init__(
        self,
        irreps_in: Optional[Dict[str, Irreps]] = None,
        embedding_dim: int = 16,
        num_species: Optional[int] = None,
        allowed_species: Optional[List[int]] = None,
        out_fields: Tuple[str, ...] = (DataKey.NODE_ATTRS, DataKey.NODE_FEATURES),
    ):
        super().__init__()
        if allowed_species is not None and num_species is not None:
            raise ValueError("allowed_species and num_species cannot both be provided.")
        self.embedding_dim = embedding_dim
        self.out_fields = tuple(out_fields)
        if allowed_species is not None:
            self.atomic_number_to_index = _AtomicNumberToIndex(allowed_species)
            num_species = self.atomic_number_to_index.num_species
        self.init_irreps(irreps_in, {k: Irreps(f"{embedding_dim}x0e") for k in self.out_fields})
        self.embedding = torch.nn.Embedding(num_species, embedding_dim)

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        if DataKey.SPECIES_INDEX in data:
            idx = data[DataKey.SPECIES_INDEX]
        elif DataKey.ATOMIC_NUMBERS in data:
            idx = self.atomic_number_to_index(data[DataKey.ATOMIC_NUMBERS])
            data[DataKey.SPECIES_INDEX] = idx
        else:
            raise ValueError("Nothing in `data` to encode. Need either species_index or atomic_numbers")
        w = self.embedding.weight
        embed = torch.nn.functional.one_hot(idx, w.shape[0]).to(w.dtype) @ w
        for k in self.out_fields:
            data[k] = embed
        if DataKey.NODE_ATTRS in self.out_fields:
            idx.tfc_num_classes = int(w.shape[0])  # lets the self connection lay out the classes without a host sync
            data[DataKey.NODE_ATTRS_CLASSES] = idx
        return data
