"""Per-edge encoders: edge vectors, spherical-harmonic edge attributes, Bessel radial
embedding.  Same classes and constructor arguments as eigenn/nn/_nequip.py; the arithmetic
runs in csrc/tfc_edge.hip (one fused pass per encoder, explicit backward so that
F = -dE/dpos comes out of a single autograd.grad)."""
from __future__ import annotations

import math
from typing import Union

import torch

from .. import keys as DataKey
from .. import ops
from ..irreps import Irreps
from ..module_irreps import ModuleIrreps

__all__ = [
    "ShiftedSoftPlus",
    "PolynomialCutoff",
    "BesselBasis",
    "SphericalHarmonicEdgeAttrs",
    "RadialBasisEdgeEncoding",
    "with_edge_vectors",
    "with_batch",
]


class ShiftedSoftPlus(torch.nn.Module):
    """eigenn/nn/_nequip.py:15-37.  Kept for API parity (e.g. as `mlp_activation`); inside the
    layer the activation is evaluated by the GEMM epilogue / gate kernel."""

    def __init__(self, beta=1, threshold=20):
        super().__init__()
        self.softplus = torch.nn.Softplus(beta=beta, threshold=threshold)
        self._log2 = math.log(2.0)

    def forward(self, x):
        return self.softplus(x) - self._log2


class PolynomialCutoff(torch.nn.Module):
    """eigenn/nn/_nequip.py:45-78: holds p and r_max; evaluated inside the radial kernel."""

    def __init__(self, r_max, p=6):
        super().__init__()
        self.register_buffer("p", torch.Tensor([p]))
        self.register_buffer("r_max", torch.Tensor([r_max]))
        self._p, self._r_max = float(p), float(r_max)


class BesselBasis(torch.nn.Module):
    """eigenn/nn/_nequip.py:82-128: trainable frequencies `bessel_weights`, init n*pi."""

    def __init__(self, r_max, num_basis=8, trainable=True):
        super().__init__()
        self.trainable = trainable
        self.num_basis = num_basis
        self.r_max = float(r_max)
        self.prefactor = 2.0 / self.r_max
        w = torch.linspace(start=1.0, end=num_basis, steps=num_basis) * math.pi
        if trainable:
            self.bessel_weights = torch.nn.Parameter(w)
        else:
            self.register_buffer("bessel_weights", w)


def with_edge_vectors(data: DataKey.Type, with_lengths: bool = True, _lmax: int = -1) -> DataKey.Type:
    """eigenn/nn/_nequip.py:214-266.  Adds EDGE_VECTORS (and EDGE_LENGTH) to the dict,
    differentiable w.r.t. positions.  (`_lmax` >= 0 is an internal hook that also returns the
    spherical harmonics computed in the same kernel pass.)"""
    if DataKey.EDGE_VECTORS in data:
        if with_lengths and DataKey.EDGE_LENGTH not in data:
            # |v| through the same kernel family: radial op needs it only as an input tensor
            data[DataKey.EDGE_LENGTH] = _vector_length(data[DataKey.EDGE_VECTORS])
        return data
    pos = data[DataKey.POSITIONS]
    ordering = ops.edge_ordering(data[DataKey.EDGE_INDEX], pos.shape[0])
    shift = cell = batch = None
    if DataKey.CELL in data:
        cell = data[DataKey.CELL].view(-1, 3, 3)
        shift = data[DataKey.EDGE_CELL_SHIFT]
        if cell.shape[0] > 1:
            batch = data[DataKey.BATCH]
    out = ops.edge_geometry(pos, ordering, shift, cell, batch, lmax=_lmax)
    data[DataKey.EDGE_VECTORS] = out[0]
    # the length always comes with the vectors (same pass); harmless when not asked for
    data[DataKey.EDGE_LENGTH] = out[1]
    if _lmax >= 0:
        data["_tfc_sh_full"] = out[2]
    return data


class _LengthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec):
        length = torch.linalg.vector_norm(vec, dim=-1)
        ctx.save_for_backward(vec, length)
        return length

    @staticmethod
    def backward(ctx, g):
        vec, length = ctx.saved_tensors
        return g[:, None] * vec / length.clamp_min(1e-12)[:, None]


def _vector_length(vec):
    # only reached when a caller supplies EDGE_VECTORS itself (not on the models' path)
    return _LengthFn.apply(vec)


def with_batch(data: DataKey.Type) -> DataKey.Type:
    """eigenn/nn/_nequip.py:270-283"""
    if DataKey.BATCH in data:
        return data
    pos = data[DataKey.POSITIONS]
    data[DataKey.BATCH] = torch.zeros(len(pos), dtype=torch.long, device=pos.device)
    return data


_SH_FACTOR = {
    "component": lambda l: 1.0,
    "integral": lambda l: 1.0 / math.sqrt(4 * math.pi),
    "norm": lambda l: 1.0 / math.sqrt(2 * l + 1),
}


class SphericalHarmonicEdgeAttrs(ModuleIrreps, torch.nn.Module):
    """eigenn/nn/_nequip.py:132-176.  EDGE_ATTRS = Y_l^m of the unit edge vector (of the edge vector
    itself with `edge_sh_normalize=False`)."""

    def __init__(
        self,
        irreps_edge_sh: Union[int, str, Irreps],
        edge_sh_normalization: str = "component",
        edge_sh_normalize: bool = True,
        irreps_in=None,
        out_field: str = DataKey.EDGE_ATTRS,
    ):
        super().__init__()
        self.out_field = out_field
        if isinstance(irreps_edge_sh, int):
            self.irreps_edge_sh = Irreps.spherical_harmonics(irreps_edge_sh)
        else:
            self.irreps_edge_sh = Irreps(irreps_edge_sh)
        for mul, ir in self.irreps_edge_sh:
            if ir.p != (-1) ** ir.l:
                raise ValueError(f"spherical harmonics irreps must have parity (-1)^l, got {self.irreps_edge_sh}")
        if edge_sh_normalization not in _SH_FACTOR:
            raise ValueError(f"unknown normalization {edge_sh_normalization}")
        self.normalize = bool(edge_sh_normalize)
        self.init_irreps(irreps_in=irreps_in, irreps_out={out_field: self.irreps_edge_sh})
        self.lmax = self.irreps_edge_sh.lmax
        if self.lmax > 3:
            raise NotImplementedError("matten_amd: spherical harmonics are built for l <= 3")
        self.normalization = edge_sh_normalization
        # the kernel emits l = 0..lmax once each, component-normalised; anything else is a
        # column selection / per-l rescale of that
        cols, scale = [], []
        for mul, ir in self.irreps_edge_sh:
            for _ in range(mul):
                cols += list(range(ir.l * ir.l, (ir.l + 1) ** 2))
                scale += [_SH_FACTOR[edge_sh_normalization](ir.l)] * ir.dim
        self._native = cols == list(range((self.lmax + 1) ** 2)) and edge_sh_normalization == "component"
        self.register_buffer("_cols", torch.tensor(cols, dtype=torch.long), persistent=False)
        self.register_buffer("_scale", torch.tensor(scale, dtype=torch.float32), persistent=False)

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        if not self.normalize and DataKey.EDGE_VECTORS not in data:
            data = with_edge_vectors(data, with_lengths=False)
        if DataKey.EDGE_VECTORS in data:
            sh = ops.spherical_harmonics(data[DataKey.EDGE_VECTORS], self.lmax, self.normalize)
        else:
            data = with_edge_vectors(data, with_lengths=False, _lmax=self.lmax)
            sh = data.pop("_tfc_sh_full")
        if not self._native:
            sh = sh.index_select(1, self._cols) * self._scale
        data[self.out_field] = sh
        return data


class RadialBasisEdgeEncoding(ModuleIrreps, torch.nn.Module):
    """eigenn/nn/_nequip.py:180-210.  EDGE_EMBEDDING = Bessel basis x polynomial cutoff."""

    def __init__(
        self,
        basis=BesselBasis,
        cutoff=PolynomialCutoff,
        basis_kwargs={},
        cutoff_kwargs={},
        out_field: str = DataKey.EDGE_EMBEDDING,
        irreps_in=None,
    ):
        super().__init__()
        self.basis = basis(**basis_kwargs)
        self.cutoff = cutoff(**cutoff_kwargs)
        if not isinstance(self.basis, BesselBasis) or not isinstance(self.cutoff, PolynomialCutoff):
            raise NotImplementedError("matten_amd: the radial kernel implements BesselBasis x PolynomialCutoff")
        self.out_field = out_field
        self.init_irreps(irreps_in=irreps_in, irreps_out={self.out_field: Irreps([(self.basis.num_basis, (0, 1))])})

    def forward(self, data: DataKey.Type) -> DataKey.Type:
        data = with_edge_vectors(data, with_lengths=True)
        # the cutoff radius of the envelope and of the basis are separate arguments in the
        # reference (both r_max in the shipped models); the kernel takes one radius
        if abs(self.basis.r_max - self.cutoff._r_max) > 1e-12:
            raise NotImplementedError("matten_amd: basis r_max and cutoff r_max must coincide")
        data[self.out_field] = ops.radial_embedding(
            data[DataKey.EDGE_VECTORS], data[DataKey.EDGE_LENGTH], self.basis.bessel_weights, self.basis.r_max, self.cutoff._p
        )
        return data
