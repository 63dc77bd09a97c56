"""`ModuleIrreps`: the plugin protocol every layer of the reference implements
(eigenn/data/irreps.py:17-165).  A layer declares the irreps of the data-dict fields it
reads (`irreps_in`) and publishes those it produces (`irreps_out` = inputs overwritten by
outputs); `create_sequential_module` (eigenn/model_factory/utils.py:52-85) chains them.
Same checks, same exceptions, so a matten_amd layer drops into that factory unchanged.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

from . import keys as DataKey
from .irreps import Irreps


def _fix_irreps_dict(d: Dict[str, object]) -> Dict[str, Optional[Irreps]]:
    return {k: (None if v is None else Irreps(v)) for k, v in d.items()}


def _check_irreps_compatible(ir1: Dict[str, Irreps], ir2: Dict[str, Irreps]) -> bool:
    return all(ir1[k] == ir2[k] for k in ir1 if k in ir2)


def _same_type(a: Irreps, b: Irreps) -> bool:
    return set(Irreps(a).ls) == set(Irreps(b).ls)


class ModuleIrreps:
    REQUIRED_KEYS_IRREPS_IN: Optional[Sequence[str]] = None
    REQUIRED_TYPE_IRREPS_IN: Optional[Dict[str, object]] = None
    OPTIONAL_MUL_TYPE_IRREPS_IN: Optional[Dict[str, object]] = None

    def init_irreps(
        self,
        irreps_in: Optional[Dict[str, object]] = None,
        irreps_out=None,
        *,
        required_keys_irreps_in: Optional[Sequence[str]] = None,
        required_type_irreps_in: Optional[Dict[str, object]] = None,
        optional_mul_type_irreps_in: Optional[Dict[str, object]] = None,
    ):
        irreps_in = self.fix_irreps_in(_fix_irreps_dict({} if irreps_in is None else irreps_in))

        if irreps_out is None:
            irreps_out = {}
        elif isinstance(irreps_out, str):
            if irreps_out not in irreps_in:
                raise AssertionError(f"`irreps_in` does not contain key for `irreps_out = {irreps_out}`")
            irreps_out = {irreps_out: irreps_in[irreps_out]}
        irreps_out = _fix_irreps_dict(irreps_out)

        required_keys = list(self.REQUIRED_KEYS_IRREPS_IN or []) + list(required_keys_irreps_in or [])
        required_type = dict(self.REQUIRED_TYPE_IRREPS_IN or {})
        required_type.update(required_type_irreps_in or {})
        required_type = _fix_irreps_dict(required_type)
        optional = dict(self.OPTIONAL_MUL_TYPE_IRREPS_IN or {})
        optional.update(optional_mul_type_irreps_in or {})
        optional = _fix_irreps_dict(optional)

        for k in required_keys + list(required_type):
            if k not in irreps_in:
                raise ValueError(f"This module {type(self)} requires `{k}` in `irreps_in`.")
        for k, v in required_type.items():
            if irreps_in[k] is None or not _same_type(irreps_in[k], v):
                raise ValueError(
                    f"This module {type(self)} expects irreps_in['{k}'] be of type {v}, instead got "
                    f"{irreps_in[k]}. Note, type means degree and parity, not multiplicity."
                )
        for k, v in optional.items():
            if k in irreps_in and irreps_in[k] != v:
                raise ValueError(f"This module {type(self)} expects irreps_in['{k}'] to be {v}, instead got {irreps_in[k]}.")

        self._irreps_in = irreps_in
        self._irreps_out = dict(irreps_in)
        self._irreps_out.update(irreps_out)

    @property
    def irreps_in(self):
        return self._irreps_in

    @property
    def irreps_out(self):
        return self._irreps_out

    def fix_irreps_in(self, irreps_in: Dict[str, Optional[Irreps]]) -> Dict[str, Optional[Irreps]]:
        irreps_in = dict(irreps_in)
        pos = DataKey.POSITIONS
        if pos in irreps_in and irreps_in[pos] != Irreps("1x1o"):
            raise ValueError(f"Positions must have irreps 1o, got `{irreps_in[pos]}`")
        irreps_in[pos] = Irreps("1o")
        ei = DataKey.EDGE_INDEX
        if ei in irreps_in and irreps_in[ei] is not None:
            raise ValueError(f"Edge indexes must have irreps `None`, got `{irreps_in[ei]}`")
        irreps_in[ei] = None
        return irreps_in
