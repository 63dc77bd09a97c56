"""`Sequential` with the irreps hand-over check (eigenn/nn/sequential.py:9-49) and the factory
helper that chains `cls(**kwargs, irreps_in=previous.irreps_out)`
(eigenn/model_factory/utils.py:13-99; its logging and DEBUG anomaly hooks are not rebuilt)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from ..irreps import Irreps
from ..module_irreps import ModuleIrreps, _check_irreps_compatible


class Sequential(torch.nn.Sequential, ModuleIrreps):
    def __init__(self, *args):
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            named = args[0]
        else:
            named = OrderedDict((f"{type(m).__name__}_{i}", m) for i, m in enumerate(args))
        mods = list(named.values())
        for i in range(len(mods) - 1):
            a, b = mods[i], mods[i + 1]
            if not _check_irreps_compatible(a.irreps_out, b.irreps_in):
                raise ValueError(
                    f"Output irreps of module {i} `{type(a).__name__}`: {a.irreps_out}` is incompatible with input irreps of "
                    f"module {i + 1} `{type(b).__name__}`: {b.irreps_in}."
                )
        torch.nn.Sequential.__init__(self, named)
        self.init_irreps(irreps_in=mods[0].irreps_in, irreps_out=mods[-1].irreps_out)


def create_sequential_module(
    modules: "OrderedDict[str, Tuple[type, Dict]]",
    irreps_in: Optional[Dict[str, Irreps]] = None,
    use_kwargs_irreps_in: bool = False,
) -> Sequential:
    built: "OrderedDict[str, torch.nn.Module]" = OrderedDict()
    last = None
    for name, (cls, kwargs) in modules.items():
        ir = irreps_in if last is None else last.irreps_out
        if "irreps_in" in kwargs:
            if not use_kwargs_irreps_in:
                raise ValueError(
                    f"Trying to automatically determine irreps_in for module {name} But it is provided as kwargs. "
                    "Set `use_kwargs_irrpes_in=True` to force it."
                )
            ir = {**ir, **kwargs["irreps_in"]} if ir is not None else kwargs["irreps_in"]  # kwargs win
        kw = {**kwargs, "irreps_in": ir}
        try:
            last = cls(**kw)
        except Exception as err:
            raise RuntimeError(f"Failed instantiate module `{cls.__name__}` with kwargs: `{kw}`") from err
        built[name] = last
    return Sequential(built)
