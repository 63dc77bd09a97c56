"""The e3nn building blocks the layer is composed of, with e3nn's parameter layout and
normalisation but computed by the libtfc kernels.

The reference constructs these from e3nn (o3.Linear, o3.FullyConnectedTensorProduct,
o3.TensorProduct, nn.FullyConnectedNet, nn.Gate; call sites eigenn/nn/point_conv.py:56-61,
130-135,160-162 and eigenn/nn/utils.py:134-140,231-252).  Parameter names, shapes and
flattening order are e3nn's, so reference state_dicts load key for key.  Each module only
derives a host-side *spec* (offsets, multiplicities, normalisation constants); all
arithmetic happens in matten_amd.ops.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch

from .. import ops
from .._lib import ACT
from ..irreps import Irrep, Irreps

# e3nn.math.normalize2mom constants: cst = E[act(z)^2]^(-1/2) over z = torch.randn(1_000_000,
# generator=manual_seed(0), dtype=float64) - a Monte-Carlo value, NOT the analytic one
# (regenerate with tools/normalize2mom_constants.py).
NORMALIZE2MOM = {
    "ssp": 1.878204668541552,
    "silu": 1.6791767923989418,
    "tanh": 1.5937334472592692,
    "abs": 1.001110600838467,
    "sigmoid": 1.8467055342154763,
    "identity": 1.0,
}

# parity behaviour of each activation (e3nn.nn.Activation): even functions map 0o -> 0e,
# odd functions keep 0o, anything else cannot act on an odd scalar.
_ACT_SYMMETRY = {"abs": "even", "tanh": "odd", "identity": "odd"}


def _n2m(name: str) -> float:
    c = NORMALIZE2MOM[name]
    return 1.0 if abs(c - 1.0) < 1e-4 else c


def _act_name(act) -> str:
    """Activation given by name, or as one of the callables the reference passes (eigenn/nn/utils.py:14-26: its
    ShiftedSoftPlus, torch.nn.functional.silu, torch.sigmoid, torch.abs, torch.tanh)."""
    if isinstance(act, str):
        name = act
    else:
        name = getattr(act, "__name__", type(act).__name__).lower()
        name = {"shiftedsoftplus": "ssp", "shifted_softplus": "ssp"}.get(name, name)
    if name not in NORMALIZE2MOM or name not in ACT:
        raise ValueError(f"matten_amd: unknown activation `{act}` (known: {sorted(set(NORMALIZE2MOM) & set(ACT))})")
    return name


def _row_blocks(irreps: Irreps):
    """(offset, mul, 2l+1) of every non-empty block of a feature row."""
    return tuple((off, m.mul, m.ir.dim) for off, m in zip(irreps.offsets(), irreps) if m.dim > 0)


class Linear(torch.nn.Module):
    """o3.Linear(irreps_in, irreps_out): one weight block [mul_in, mul_out] per pair of equal
    irreps (loop order in, out), flattened into a 1-D `weight`; path weight
    1/sqrt(sum of mul_in reaching the output block); no bias."""

    def __init__(self, irreps_in, irreps_out):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_out)
        in_offs, out_offs = self.irreps_in.offsets(), self.irreps_out.offsets()
        pairs = [
            (a, b)
            for a, mi in enumerate(self.irreps_in)
            for b, mo in enumerate(self.irreps_out)
            if mi.ir == mo.ir
        ]
        ins, w_off = [], 0
        for a, b in pairs:
            mi, mo = self.irreps_in[a], self.irreps_out[b]
            fan = sum(self.irreps_in[a2].mul for a2, b2 in pairs if b2 == b)
            if mi.mul > 0 and mo.mul > 0:
                ins.append((in_offs[a], mi.mul, out_offs[b], mo.mul, mi.ir.dim, w_off, fan**-0.5))
            w_off += mi.mul * mo.mul
        self.weight_numel = w_off
        self.weight = torch.nn.Parameter(torch.randn(w_off))
        covered_out = {b for _, b in pairs}
        covered_in = {a for a, _ in pairs}
        self.spec = ops.LinearSpec(
            d_in=self.irreps_in.dim,
            d_out=self.irreps_out.dim,
            weight_numel=w_off,
            instructions=tuple(ins),
            out_covered=all(b in covered_out or self.irreps_out[b].dim == 0 for b in range(len(self.irreps_out))),
            in_covered=all(a in covered_in or self.irreps_in[a].dim == 0 for a in range(len(self.irreps_in))),
            out_blocks=_row_blocks(self.irreps_out),
        )

    def forward(self, x):
        return ops.linear(x, self.weight, self.spec)

    def extra_repr(self):
        return f"{self.irreps_in} -> {self.irreps_out} | {self.weight_numel} weights"


class FullyConnectedTensorProduct(torch.nn.Module):
    """o3.FullyConnectedTensorProduct(in1, in2, out) with internal shared weights: 'uvw'
    instructions for every (i1, i2, io) with ir_out in ir1 x ir2 (loop order i1, i2, io),
    weight blocks [mul1, mul2, mul_out] flattened into one `weight`.

    The fast route is the case the layer uses - a scalar (l = 0) second input, i.e. the species-embedding node
    attributes (eigenn/nn/point_conv.py:159-164): one grouped MFMA product (ops.fctp).  Any other second input (the
    reference's tests/nn/test_convolution.py gives the node attributes 4x1e+2x0e) takes the general route, built from
    the same kernels in two steps: the Clebsch-Gordan contraction z[n, (v, u), k] = sum_ij C[i,j,k] x[n,u,i] a[n,v,j]
    is the 'uvu' convolution kernel on an ordering in which every node is its own sender and receiver (one unit-weight
    path per copy v of the second input), the channel mixing out[n,w,k] = sqrt(alpha) sum_{(v,u)} W[u,v,w] z[n,(v,u),k]
    is an o3.Linear product.  Not tuned: it exists so that everything the reference can construct also runs."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1).simplify()
        self.irreps_in2 = Irreps(irreps_in2).simplify()
        self.irreps_out = Irreps(irreps_out).simplify()
        o1, o2, oo = self.irreps_in1.offsets(), self.irreps_in2.offsets(), self.irreps_out.offsets()
        triples = [
            (a, b, c)
            for a, m1 in enumerate(self.irreps_in1)
            for b, m2 in enumerate(self.irreps_in2)
            for c, mo in enumerate(self.irreps_out)
            if mo.ir in m1.ir * m2.ir
        ]
        ins, w_off = [], 0
        self._unsupported = None
        self._general = any(self.irreps_in2[b].ir.l != 0 for _, b, _ in triples)
        g_paths, g_lin, g_blocks, z_off, zw_off, lw_off = [], [], [], 0, 0, 0
        for a, b, c in triples:
            m1, m2, mo = self.irreps_in1[a], self.irreps_in2[b], self.irreps_out[c]
            fan = sum(self.irreps_in1[a2].mul * self.irreps_in2[b2].mul for a2, b2, c2 in triples if c2 == c)
            alpha = mo.ir.dim / fan if fan > 0 else 0.0
            if m1.mul * m2.mul * mo.mul > 0:
                if self._general:
                    if max(m1.ir.l, m2.ir.l, mo.ir.l) > 3:
                        self._unsupported = "l > 3"
                    g_lin.append((z_off, m2.mul * m1.mul, oo[c], mo.mul, mo.ir.dim, lw_off, math.sqrt(alpha)))
                    g_blocks.append((w_off, m1.mul, m2.mul, mo.mul))
                    for v in range(m2.mul):
                        g_paths.append(dict(l1=m1.ir.l, l2=m2.ir.l, l3=mo.ir.l, mul=m1.mul, x_off=o1[a], y_off=o2[b] + v * m2.ir.dim,
                                            out_off=z_off, w_off=zw_off, coeff=1.0))
                        z_off += m1.mul * mo.ir.dim
                        zw_off += m1.mul
                    lw_off += m1.mul * m2.mul * mo.mul
                else:
                    # C(l,0,l)[i,0,k] = delta_ik / sqrt(2l+1)
                    coeff = math.sqrt(alpha) / math.sqrt(mo.ir.dim)
                    ins.append((o1[a], m1.mul, o2[b], m2.mul, oo[c], mo.mul, mo.ir.dim, w_off, coeff))
            w_off += m1.mul * m2.mul * mo.mul
        self.weight_numel = w_off
        self.weight = torch.nn.Parameter(torch.randn(w_off))
        cov = lambda irreps, idx: all(i in idx or irreps[i].dim == 0 for i in range(len(irreps)))
        if self._general and self._unsupported is None:
            self._g_blocks, self._g_w = tuple(g_blocks), zw_off
            self._g_plan = ops.ConvPlan(g_paths, self.irreps_in1.dim, self.irreps_in2.dim, z_off, zw_off)
            self._g_lin = ops.LinearSpec(
                d_in=z_off, d_out=self.irreps_out.dim, weight_numel=lw_off, instructions=tuple(g_lin),
                out_covered=cov(self.irreps_out, {c for _, _, c in triples}), in_covered=True, out_blocks=_row_blocks(self.irreps_out),
            )
        self.spec = ops.FctpSpec(
            d_x=self.irreps_in1.dim,
            d_a=self.irreps_in2.dim,
            d_out=self.irreps_out.dim,
            weight_numel=w_off,
            instructions=tuple(ins),
            out_covered=cov(self.irreps_out, {c for _, _, c in triples}),
            x_covered=cov(self.irreps_in1, {a for a, _, _ in triples}),
            a_covered=cov(self.irreps_in2, {b for _, b, _ in triples}),
            out_blocks=_row_blocks(self.irreps_out),
        )

    def forward(self, x, a, base: Optional[torch.Tensor] = None, classes: Optional[torch.Tensor] = None):
        if self._unsupported:
            raise NotImplementedError(f"matten_amd FullyConnectedTensorProduct: {self._unsupported}")
        if self._general:
            n = x.shape[0]
            ordering = ops.self_ordering(n, x.device)
            ones = torch.ones(n, self._g_w, dtype=torch.float32, device=x.device)
            z = ops.conv(x, a, ones, self._g_plan, ordering, 1.0, False)
            # e3nn keeps a block as [u, v, w]; the Linear contracts over (v, u), the layout the convolution wrote
            wl = torch.cat([self.weight[o : o + m1 * m2 * mo].view(m1, m2, mo).permute(1, 0, 2).reshape(-1) for o, m1, m2, mo in self._g_blocks])
            out = ops.linear(z, wl, self._g_lin)
            return out if base is None else out + base
        return ops.fctp(x, a, self.weight, self.spec, base, classes=classes)

    def extra_repr(self):
        return f"{self.irreps_in1} x {self.irreps_in2} -> {self.irreps_out} | {self.weight_numel} weights"


class _FcLayer(torch.nn.Module):
    def __init__(self, h_in: int, h_out: int):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(h_in, h_out))


class FullyConnectedNet(torch.nn.Module):
    """e3nn.nn.FullyConnectedNet(hs, act): layers `layer{i}.weight` [h_in, h_out] ~ randn, no
    bias; hidden layers x <- cst*act(x @ W / sqrt(h_in)), last layer x <- x @ W / sqrt(h_in)."""

    def __init__(self, hs: Sequence[int], act: str = "ssp"):
        super().__init__()
        self.hs = [int(h) for h in hs]
        act = _act_name(act)
        self.act = act
        self.act_cst = _n2m(act)
        for i, (a, b) in enumerate(zip(self.hs, self.hs[1:])):
            setattr(self, f"layer{i}", _FcLayer(a, b))

    def _weights(self):
        return [getattr(self, f"layer{i}").weight for i in range(len(self.hs) - 1)]

    def forward(self, x, pre=None):
        """`pre`: this network's activations for `x`, from precompute() (see ops.radial_mlp)."""
        if self.act != "ssp":
            # the fused kernels are written for the shifted softplus (the only activation the two target models use,
            # eigenn/nn/utils.py:182); the others run layer by layer on the dense product and the gate kernel
            return ops.radial_mlp_general(x, self._weights(), self.act_cst, ACT[self.act])
        return ops.radial_mlp(x, self._weights(), self.act_cst, pre)

    def precompute(self, x):
        if self.act != "ssp":
            return None
        return ops.radial_mlp_precompute(x, self._weights(), self.act_cst)

    def extra_repr(self):
        return f"{self.hs} act={self.act}"


class Gate(torch.nn.Module):
    """e3nn.nn.Gate(irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated) with the
    activations given by name.  Input [scalars | gates | gated] (irreps_in is that sum,
    simplified), output [act(scalars) | act(gates) * gated]."""

    def __init__(self, irreps_scalars, act_scalars: Sequence[str], irreps_gates, act_gates: Sequence[str], irreps_gated):
        super().__init__()
        self.irreps_scalars = Irreps(irreps_scalars)
        self.irreps_gates = Irreps(irreps_gates)
        self.irreps_gated = Irreps(irreps_gated)
        if len(self.irreps_gates) > 0 and self.irreps_gates.lmax > 0:
            raise ValueError(f"Gate scalars must be scalars, instead got irreps_gates = {self.irreps_gates}")
        if len(self.irreps_scalars) > 0 and self.irreps_scalars.lmax > 0:
            raise ValueError(f"Scalars must be scalars, instead got irreps_scalars = {self.irreps_scalars}")
        if self.irreps_gates.num_irreps != self.irreps_gated.num_irreps:
            raise ValueError(
                f"There are {self.irreps_gated.num_irreps} irreps in irreps_gated, but a different number "
                f"({self.irreps_gates.num_irreps}) of gate scalars in irreps_gates"
            )
        if len(act_scalars) != len(self.irreps_scalars) or len(act_gates) != len(self.irreps_gates):
            raise ValueError("Gate: one activation per irreps block is required")
        self.irreps_in = (self.irreps_scalars + self.irreps_gates + self.irreps_gated).simplify()

        def out_parity(act, p_in):
            if p_in == 1:
                return 1
            sym = _ACT_SYMMETRY.get(act)
            if sym == "even":
                return 1
            if sym == "odd":
                return -1
            raise ValueError(f"activation `{act}` cannot act on an odd scalar: it is neither even nor odd")

        segs = []
        out_irreps: List = []
        in_off = out_off = 0
        for (mul, ir), act in zip(self.irreps_scalars, act_scalars):
            segs.append((in_off, out_off, mul, 1, -1, ACT[act], _n2m(act)))
            out_irreps.append((mul, Irrep(0, out_parity(act, ir.p))))
            in_off += mul
            out_off += mul
        # one (activation, parity) per gate channel, in order
        gate_info = []
        g_off = in_off
        for (mul, ir), act in zip(self.irreps_gates, act_gates):
            p_out = out_parity(act, ir.p)
            gate_info += [(act, p_out)] * mul
        in_off += self.irreps_gates.dim
        u = 0
        for mul, ir in self.irreps_gated.simplify():
            start = 0
            while start < mul:  # split where the gate's activation / parity changes
                act, p = gate_info[u + start]
                end = start
                while end < mul and gate_info[u + end] == (act, p):
                    end += 1
                cnt = end - start
                segs.append((in_off + start * ir.dim, out_off + start * ir.dim, cnt, ir.dim, g_off + u + start, ACT[act], _n2m(act)))
                out_irreps.append((cnt, Irrep(ir.l, ir.p * p)))
                start = end
            in_off += mul * ir.dim
            out_off += mul * ir.dim
            u += mul
        self.irreps_out = Irreps(out_irreps)
        self.spec = ops.GateSpec(d_in=self.irreps_in.dim, d_out=self.irreps_out.dim, segs=tuple(segs),
                                 in_blocks=_row_blocks(self.irreps_in), out_blocks=_row_blocks(self.irreps_out))

    def forward(self, x):
        return ops.gate(x, self.spec)

    def extra_repr(self):
        return f"{self.irreps_in} -> {self.irreps_out}"


class UVUTensorProductCore(torch.nn.Module):
    """o3.TensorProduct(in1, in2, mid, instructions 'uvu', shared_weights=False,
    internal_weights=False) fused with the gather of in1 rows and the per-node sum.

    `instructions` = [(i_in1, i_in2, i_out)] in weight-slot order (eigenn/nn/utils.py:205-229).
    Normalisation: irrep_normalization='component', path_normalization='element' =>
    coefficient sqrt((2 l3 + 1) / sum of mul_in2 over the instructions sharing the output)."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        self.instructions = [tuple(i[:3]) for i in instructions]
        o1, o2, oo = self.irreps_in1.offsets(), self.irreps_in2.offsets(), self.irreps_out.offsets()
        paths, w_off = [], 0
        self._unsupported = None
        for i1, i2, io in self.instructions:
            m1, m2, mo = self.irreps_in1[i1], self.irreps_in2[i2], self.irreps_out[io]
            if mo.ir not in m1.ir * m2.ir:
                raise ValueError(f"{m1.ir} x {m2.ir} does not contain {mo.ir}")
            if m1.mul != mo.mul:
                raise ValueError("'uvu' needs mul_in1 == mul_out")
            fan = sum(self.irreps_in2[j2].mul for _, j2, jo in self.instructions if jo == io)
            if max(m1.ir.l, m2.ir.l, mo.ir.l) > 3:
                self._unsupported = "l > 3"
            paths.append(
                dict(l1=m1.ir.l, l2=m2.ir.l, l3=mo.ir.l, mul=m1.mul, x_off=o1[i1], y_off=o2[i2], out_off=oo[io],
                     w_off=w_off, coeff=math.sqrt(mo.ir.dim / fan), mul2=m2.mul, io=io)
            )
            w_off += m1.mul * m2.mul
        self.weight_numel = w_off
        # Second inputs with blocks of multiplicity > 1 (not spherical harmonics: the reference's own constructor test
        # gives the edge attributes 4x1e+2x0e): the weight block of a path is [mul_1, mul_2] and the path sums over the
        # copies v.  The kernel's paths have one copy, so the instruction is laid out as mul_2 paths - copy v of the block
        # is itself a block of multiplicity 1 at y_off + v (2 l2 + 1) - each with a slot of its own in a wider output
        # row, the per-edge weights are re-ordered to [path, v][u] and the slots of a path are added up afterwards.
        self._expanded = self._unsupported is None and any(p["mul2"] != 1 for p in paths)
        if self._unsupported is not None:
            self.plan = None
        elif not self._expanded:
            self.plan = ops.ConvPlan([{k: v for k, v in p.items() if k not in ("mul2", "io")} for p in paths if p["mul"] > 0],
                                     self.irreps_in1.dim, self.irreps_in2.dim, self.irreps_out.dim, w_off,
                                     out_blocks=_row_blocks(self.irreps_out.simplify()))  # the blocks of the o3.Linear that follows
        else:
            ex, perm, slots, e_off, ew_off = [], [], [], 0, 0
            for p in paths:
                if p["mul"] == 0:
                    continue
                blk = p["mul"] * (2 * p["l3"] + 1)
                slots.append((p["io"], e_off, p["mul2"], blk))
                for v in range(p["mul2"]):
                    ex.append(dict(l1=p["l1"], l2=p["l2"], l3=p["l3"], mul=p["mul"], x_off=p["x_off"], y_off=p["y_off"] + v * (2 * p["l2"] + 1),
                                   out_off=e_off, w_off=ew_off, coeff=p["coeff"]))
                    perm += [p["w_off"] + u * p["mul2"] + v for u in range(p["mul"])]
                    e_off += blk
                    ew_off += p["mul"]
            self.plan = ops.ConvPlan(ex, self.irreps_in1.dim, self.irreps_in2.dim, e_off, ew_off)
            self.register_buffer("_w_perm", torch.tensor(perm, dtype=torch.int64), persistent=False)
            self._slots = tuple(slots)

    def forward(self, x, y, weight, ordering: ops.EdgeOrdering, scale: float = 1.0, per_edge: bool = False):
        if self._unsupported:
            raise NotImplementedError(f"matten_amd uvu tensor product: {self._unsupported}")
        if self._expanded:
            wide = ops.conv(x, y, weight.index_select(1, self._w_perm), self.plan, ordering, scale, per_edge)
            n = wide.shape[0]
            parts = []
            for io, m in enumerate(self.irreps_out):
                acc = None
                for jo, e_off, mul2, blk in self._slots:
                    if jo == io:
                        t = wide[:, e_off : e_off + mul2 * blk].reshape(n, mul2, blk).sum(1)
                        acc = t if acc is None else acc + t
                parts.append(acc if acc is not None else wide.new_zeros(n, m.dim))
            return torch.cat(parts, dim=1)
        return ops.conv(x, y, weight, self.plan, ordering, scale, per_edge)

    def extra_repr(self):
        return f"{self.irreps_in1} x {self.irreps_in2} -> {self.irreps_out} | {len(self.instructions)} paths, {self.weight_numel} weights/edge"


class NormActivation(torch.nn.Module):
    """e3nn.nn.NormActivation(irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False)
    with the non-linearity given by name: every irrep copy is scaled by act(|x|) (divided by |x| when
    `normalize`), |x| clamped from below at epsilon (a clamped copy passes no gradient through |x|).
    Same irreps in and out.  The activation is applied as is (no second-moment rescaling), as e3nn
    does here.  `bias=True` (a learnable shift of the norms) is not built: the reference passes
    bias=False (eigenn/nn/utils.py:142-150)."""

    def __init__(self, irreps_in, scalar_nonlinearity: str, normalize: bool = True, epsilon: Optional[float] = None, bias: bool = False):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_in)
        if epsilon is None and normalize:
            epsilon = 1e-8  # e3nn's default when normalising
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        if bias:
            raise NotImplementedError("matten_amd NormActivation: bias=True is not built (the reference uses bias=False)")
        if scalar_nonlinearity not in ACT:
            raise ValueError(f"unknown activation `{scalar_nonlinearity}`")
        self.scalar_nonlinearity = scalar_nonlinearity
        self.normalize = bool(normalize)
        self.epsilon = epsilon
        self._eps_squared = 0.0 if epsilon is None else float(epsilon) * float(epsilon)
        segs, off = [], 0
        for mul, ir in self.irreps_in:
            if mul > 0:
                segs.append((off, off, mul, ir.dim, -1, ACT[scalar_nonlinearity], 1.0))
            off += mul * ir.dim
        self.spec = ops.GateSpec(d_in=self.irreps_in.dim, d_out=self.irreps_in.dim, segs=tuple(segs))

    def forward(self, x):
        return ops.norm_activation(x, self.spec, self._eps_squared, self.normalize)

    def extra_repr(self):
        return f"{self.irreps_in} | {self.scalar_nonlinearity}, normalize={self.normalize}, epsilon={self.epsilon}"
