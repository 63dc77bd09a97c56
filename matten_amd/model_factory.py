"""The two target backbones, wired as the reference's factories wire them:
`create_energy_model` (eigenn/model_factory/nequip_energy_model.py:57-208, scripts/configs/minimal.yaml)
and `create_atomic_tensor_model` (eigenn/model_factory/atomic_tensor_model.py:50-228,
scripts/configs/minimal_atomic.yaml).  Same hparams keys, same layer names (hence the same
state_dict keys), same output fields.  The Lightning wrapper around them is out of scope."""
from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict

from . import keys as DataKey
from .nn.edge_encoding import RadialBasisEdgeEncoding, SphericalHarmonicEdgeAttrs
from .nn.embedding import SpeciesEmbedding
from .nn.nodewise import NodewiseLinear, NodewiseReduce, NodewiseSelect
from .nn.point_conv import PointConvMessagePassing
from .nn.readout import CartesianTensor, IrrepsToCartesianTensor
from .nn.sequential import create_sequential_module

OUT_FIELD_ENERGY = DataKey.TOTAL_ENERGY
OUT_FIELD_ATOM = "atomic_output"
OUT_FIELD_SELECTED = "tensor_output"
NODE_MASKS = "node_masks"


def _encoder_and_conv_layers(hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]) -> "OrderedDict[str, tuple]":
    r_cut = hparams["radial_basis_r_cut"]
    layers: "OrderedDict[str, tuple]" = OrderedDict()
    layers["one_hot"] = (SpeciesEmbedding, {"embedding_dim": hparams["species_embedding_dim"], "allowed_species": dataset_hparams["allowed_species"]})
    layers["spharm_edges"] = (SphericalHarmonicEdgeAttrs, {"irreps_edge_sh": hparams["irreps_edge_sh"]})
    layers["radial_basis"] = (RadialBasisEdgeEncoding, {
        "basis_kwargs": {"num_basis": hparams["num_radial_basis"], "r_max": r_cut},
        "cutoff_kwargs": {"r_max": r_cut},
    })
    avg = hparams["average_num_neighbors"]
    if isinstance(avg, str) and avg.lower() == "auto":
        avg = dataset_hparams["average_num_neighbors"]
    for i in range(hparams["num_layers"]):
        layers[f"layer{i}_convnet"] = (PointConvMessagePassing, {
            "conv_layer_irreps": hparams["conv_layer_irreps"],
            "activation_type": hparams["nonlinearity_type"],
            "use_resnet": hparams["resnet"],
            "message_kwargs": {"fc_num_hidden_layers": hparams["invariant_layers"], "fc_hidden_size": hparams["invariant_neurons"]},
            "update_kwargs": {"use_self_connection": hparams["use_sc"], "avg_num_neighbors": avg},
        })
    return layers


def create_energy_model(hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]):
    """... -> NodewiseLinear(hidden) -> NodewiseLinear(1x0e -> atomic_energy) -> NodewiseReduce -> total_energy"""
    layers = _encoder_and_conv_layers(hparams, dataset_hparams)
    layers["conv_to_output_hidden"] = (NodewiseLinear, {"irreps_out": hparams["conv_to_output_hidden_irreps_out"]})
    layers["output_hidden_to_scalar"] = (NodewiseLinear, {"irreps_out": "1x0e", "out_field": DataKey.PER_ATOM_ENERGY})
    reduce = hparams["reduce"]
    layers[f"total_energy_{reduce}"] = (NodewiseReduce, {"reduce": reduce, "field": DataKey.PER_ATOM_ENERGY, "out_field": OUT_FIELD_ENERGY})
    return create_sequential_module(layers, use_kwargs_irreps_in=True)


def create_atomic_tensor_model(hparams: Dict[str, Any], dataset_hparams: Dict[str, Any]):
    """... -> NodewiseLinear(hidden) -> NodewiseLinear(tensor irreps) -> NodewiseSelect(node_masks)
    [-> IrrepsToCartesianTensor when output_format == "cartesian"]; result in `tensor_output`."""
    layers = _encoder_and_conv_layers(hparams, dataset_hparams)
    fmt = hparams["output_format"]
    if fmt == "cartesian":
        out_irreps = CartesianTensor(hparams["output_formula"])
    elif fmt == "irreps":
        out_irreps = hparams["output_formula"]
    else:
        raise ValueError(f"Expect `output_format` to be one of ['cartesian', 'irreps']; got {fmt}")
    layers["conv_to_output_hidden"] = (NodewiseLinear, {"irreps_out": hparams["conv_to_output_hidden_irreps_out"]})
    layers["output_hidden_to_tensor"] = (NodewiseLinear, {"irreps_out": out_irreps, "out_field": OUT_FIELD_ATOM})
    layers["output_irreps_tensor"] = (NodewiseSelect, {"field": OUT_FIELD_ATOM, "out_field": OUT_FIELD_SELECTED, "mask_field": NODE_MASKS})
    if fmt == "cartesian":
        layers["output_cartesian_tensor"] = (IrrepsToCartesianTensor, {"formula": hparams["output_formula"], "field": OUT_FIELD_SELECTED})
    return create_sequential_module(layers, use_kwargs_irreps_in=True)
