"""`ActivationLayer`, `UVUTensorProduct`, `scatter_add`, `tp_path_exists`: the helpers of
eigenn/nn/utils.py that sit on the tensor-field-convolution path, same constructors and
irreps logic, kernels from libtfc."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from ..irreps import Irrep, Irreps, tp_path_exists
from .e3 import FullyConnectedNet, Gate, NormActivation, UVUTensorProductCore

__all__ = ["ACTIVATION", "ActivationLayer", "UVUTensorProduct", "scatter_add", "tp_path_exists"]

# eigenn/nn/utils.py:14-26 - names only; the functions themselves live in csrc/tfc_gate.hip
ACTIVATION = {"e": {"ssp": "ssp", "silu": "silu", "sigmoid": "sigmoid"}, "o": {"abs": "abs", "tanh": "tanh"}}


class ActivationLayer(torch.nn.Module):
    """eigenn/nn/utils.py:29-168.  Decides which irreps the convolution must produce so that a
    Gate can act on them: scalars and gated tensors are the requested irreps reachable by
    in1 x in2; every gated irrep gets one gate scalar, `0e` if reachable, else `0o`."""

    def __init__(
        self,
        tp_irreps_in1,
        tp_irreps_in2,
        tp_irreps_out,
        *,
        activation_type: str = "gate",
        activation_scalars: Optional[Dict[str, str]] = None,
        activation_gates: Optional[Dict[str, str]] = None,
    ):
        super().__init__()
        key = {"e": 1, "o": -1}
        if activation_scalars is None:
            act_s = {1: "ssp", -1: "tanh"}
        else:
            act_s = {key[k]: ACTIVATION[k][v] for k, v in activation_scalars.items()}
        if activation_gates is None:
            act_g = {1: "ssp", -1: "abs"}
        else:
            act_g = {key[k]: ACTIVATION[k][v] for k, v in activation_gates.items()}

        tp_irreps_out = Irreps(tp_irreps_out).sort()[0].simplify()
        irreps_scalars = Irreps(
            [(mul, ir) for mul, ir in tp_irreps_out if ir.l == 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
        )
        irreps_gated = Irreps(
            [(mul, ir) for mul, ir in tp_irreps_out if ir.l > 0 and tp_path_exists(tp_irreps_in1, tp_irreps_in2, ir)]
        )
        if activation_type == "gate":
            if irreps_gated.dim > 0:
                if tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0e"):
                    ir = "0e"
                elif tp_path_exists(tp_irreps_in1, tp_irreps_in2, "0o"):
                    ir = "0o"
                else:
                    raise ValueError(
                        f"tp_irreps_in1={tp_irreps_in1} times tp_irreps_in2={tp_irreps_in2} is unable to produce "
                        f"gates needed for irreps_gated={irreps_gated}"
                    )
            else:
                ir = None
            irreps_gates = Irreps([(mul, ir) for mul, _ in irreps_gated]).simplify()
            self.activation = Gate(
                irreps_scalars=irreps_scalars,
                act_scalars=[act_s[ir.p] for _, ir in irreps_scalars],
                irreps_gates=irreps_gates,
                act_gates=[act_g[ir.p] for _, ir in irreps_gates],
                irreps_gated=irreps_gated,
            )
        elif activation_type == "norm":
            self.activation = NormActivation(
                irreps_in=(irreps_scalars + irreps_gated).simplify(),
                scalar_nonlinearity=act_s[1],  # the norm is an even scalar (eigenn/nn/utils.py:146-147)
                normalize=True,
                epsilon=1e-8,
                bias=False,
            )
        else:
            raise ValueError(f"Support `activation_type` includes ('gate', 'norm'), got {activation_type}")

    def forward(self, x):
        return self.activation(x)

    @property
    def irreps_in(self):
        return self.activation.irreps_in

    @property
    def irreps_out(self):
        return self.activation.irreps_out


class UVUTensorProduct(torch.nn.Module):
    """eigenn/nn/utils.py:171-279.  Builds the 'uvu' instruction list, sorts the intermediate
    irreps, and owns the radial network `weight_nn` whose output weights every path per edge.

    forward(data1, data2, data_weight) keeps the reference signature, with two extra keyword
    arguments that let the caller fuse what precedes and follows the product in the reference:
    `ordering` (when given, `data1` holds NODE rows and the sender gather and the sum over each
    receiver's edges happen inside the kernel) and `scale`."""

    def __init__(
        self,
        irreps_in1,
        irreps_in2,
        irreps_out,
        *,
        internal_and_share_weights: bool = False,
        mlp_input_size: Optional[int] = None,
        mlp_hidden_size: int = 8,
        mlp_num_hidden_layers: int = 1,
        mlp_activation: str = "ssp",
    ):
        super().__init__()
        irreps_in1, irreps_in2, irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        irreps_mid, instructions = [], []
        for i, (mul, ir_in1) in enumerate(irreps_in1):
            for j, (_, ir_in2) in enumerate(irreps_in2):
                for ir_out in ir_in1 * ir_in2:
                    # the reference's second clause `or ir_out == Irreps("0e")` (utils.py:211) compares
                    # an Irrep with an Irreps and is never true
                    if ir_out in irreps_out:
                        instructions.append((i, j, len(irreps_mid)))
                        irreps_mid.append((mul, ir_out))
        irreps_mid = Irreps(irreps_mid)
        assert irreps_mid.dim > 0, (
            f"irreps_in1={irreps_in1} times irreps_in2={irreps_in2} produces no instructions in irreps_out={irreps_out}"
        )
        self.irreps_mid, permutation, _ = irreps_mid.sort()
        instructions = [(i1, i2, permutation[io]) for i1, i2, io in instructions]
        self.tp = UVUTensorProductCore(irreps_in1, irreps_in2, self.irreps_mid, instructions)

        if internal_and_share_weights:
            self.weight_nn = None
            self.weight = torch.nn.Parameter(torch.randn(self.tp.weight_numel))
        else:
            assert mlp_input_size is not None, (
                "Expect `mlp_input_size` be provided when `internal_and_share_weights` is set to `False`, got `None`"
            )
            act = mlp_activation  # a name, or one of the reference's callables (nn/e3.py:_act_name)
            layer_sizes = [mlp_input_size] + mlp_num_hidden_layers * [mlp_hidden_size] + [self.tp.weight_numel]
            self.weight_nn = FullyConnectedNet(layer_sizes, act=act)

    def forward(self, data1, data2, data_weight=None, *, ordering: Optional[ops.EdgeOrdering] = None, scale: float = 1.0,
                weight: Optional[torch.Tensor] = None):
        """`weight`: per-edge weights already computed by `self.weight_nn` (the caller ran it elsewhere,
        e.g. on a side stream); otherwise they are computed here from `data_weight`, as in the reference."""
        if weight is not None:
            pass
        elif self.weight_nn is not None:
            assert data_weight is not None, "data for weight not provided"
            weight = self.weight_nn(data_weight)
        else:
            weight = self.weight.unsqueeze(0).expand(data2.shape[0], -1).contiguous()
        if ordering is not None:
            return self.tp(data1, data2, weight, ordering, scale, per_edge=False)
        # unfused call: data1 rows are per edge already -> each edge reads its own row
        e = data2.shape[0]
        ident = torch.arange(e, dtype=torch.int64, device=data2.device)
        own = ops.EdgeOrdering(torch.stack([ident, ident]), e)
        return self.tp(data1, data2, weight, own, scale, per_edge=True)

    @property
    def irreps_out(self):
        return self.irreps_mid.simplify()


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim_size: int) -> torch.Tensor:
    """eigenn/nn/utils.py:351-357.  Only used when a caller insists on the unfused
    EDGE_MESSAGE -> scatter route; the layer itself sums inside the convolution kernel."""
    return ops.scatter_add(src, index, dim_size)
