"""Node-wise tail of the two target models (eigenn/nn/nodewise.py): a linear map on one field of
every node, a per-graph reduction, a row selection.  The linear map is the same o3.Linear the conv
layers use (fp32 MFMA, matten_amd.ops.linear); sum / mean reduce with the fixed-order segment
sum of the unfused scatter route (tfc_segment_sum)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import keys as DataKey
from .. import ops
from ..irreps import Irreps
from ..module_irreps import ModuleIrreps
from .e3 import Linear
from .edge_encoding import with_batch


class NodewiseSelect(ModuleIrreps, torch.nn.Module):
    """nodewise.py:18-87.  `data[mask_field]` is a boolean [N]; None selects every node.  Works on a
    shallow copy of the dict, like the reference."""

    def __init__(self, irreps_in: Dict[str, Irreps], field: str = DataKey.NODE_FEATURES, out_field: Optional[str] = None,
                 mask_field: Optional[str] = None):
        super().__init__()
        self.field = field
        self.out_field = field if out_field is None else out_field
        self.mask_field = mask_field
        self.init_irreps(irreps_in, {self.out_field: irreps_in.get(field)}, required_keys_irreps_in=[field])

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data = data.copy()
        value = data[self.field]
        data[self.out_field] = value if self.mask_field is None else value[data[self.mask_field]]
        return data

    def extra_repr(self):
        return f"{self.field} -> {self.out_field} ({self.irreps_out[self.out_field]}), mask: {self.mask_field}"


class NodewiseLinear(ModuleIrreps, torch.nn.Module):
    """nodewise.py:90-118.  state_dict key: `linear.weight` (e3nn's flat layout)."""

    def __init__(self, irreps_in: Dict[str, Irreps], irreps_out=None, field: str = DataKey.NODE_FEATURES,
                 out_field: Optional[str] = None):
        super().__init__()
        self.field = field
        self.out_field = field if out_field is None else out_field
        self.init_irreps(irreps_in, {self.out_field: irreps_in.get(field) if irreps_out is None else irreps_out},
                         required_keys_irreps_in=[field])
        self.linear = Linear(self.irreps_in[field], self.irreps_out[self.out_field])

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data[self.out_field] = self.linear(data[self.field])
        return data


class NodewiseReduce(ModuleIrreps, torch.nn.Module):
    """nodewise.py:121-149: per-graph reduction of a node field over `batch`.  The number of graphs
    is taken from `data["ptr"]` (PyG collate) when present, else from batch.max() (a host sync, as
    in torch_scatter)."""

    def __init__(self, irreps_in: Dict[str, Irreps], field: str, out_field: Optional[str] = None, reduce: str = "sum"):
        super().__init__()
        assert reduce in ("sum", "mean", "min", "max")
        self.reduce = reduce
        self.field = field
        self.out_field = f"{reduce}_{field}" if out_field is None else out_field
        self.init_irreps(irreps_in, {self.out_field: irreps_in.get(field)}, required_keys_irreps_in=[field])

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        with_batch(data)
        x, batch = data[self.field], data[DataKey.BATCH]
        n_graphs = int(data["ptr"].numel()) - 1 if "ptr" in data else int(batch.max()) + 1
        if self.reduce in ("sum", "mean"):
            out = ops.scatter_add(x, batch, n_graphs)
            if self.reduce == "mean":
                out = out / torch.bincount(batch, minlength=n_graphs).clamp_min(1).to(out.dtype)[:, None]
        else:  # order statistics have no sum to keep in order: torch's device scatter_reduce
            if not x.is_cuda:
                raise RuntimeError("matten_amd: tensors must live on a ROCm device (no CPU fallback exists)")
            out = torch.zeros(n_graphs, x.shape[1], dtype=x.dtype, device=x.device).scatter_reduce_(
                0, batch.view(-1, 1).expand_as(x), x, reduce="amin" if self.reduce == "min" else "amax", include_self=False)
        data[self.out_field] = out
        return data
