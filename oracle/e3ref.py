"""Oracle (test infrastructure): restatement of the e3nn semantics the hot path relies on.

e3nn is NOT vendored in /root/reference and is not importable here (setup.py:34 lists a
bare, unpinned "e3nn"; era-consistent version ~0.4.x).  This file restates its published
algorithm for exactly the pieces the reference calls:

  * o3.Irrep / o3.Irreps algebra            (used all over eigenn/nn/utils.py, point_conv.py)
  * o3.wigner_3j                            (real Clebsch-Gordan, unit Frobenius norm)
  * o3.SphericalHarmonics                   (eigenn/nn/_nequip.py:167-169)
  * o3.TensorProduct ('uvu','uvw','uuu')    (eigenn/nn/utils.py:231-238)
  * o3.Linear, o3.FullyConnectedTensorProduct (eigenn/nn/point_conv.py:56-61,130-135,160-162)
  * nn.FullyConnectedNet + normalize2mom    (eigenn/nn/utils.py:247-252)
  * nn.Gate (Activation + ElementwiseTensorProduct) (eigenn/nn/utils.py:134-140)
  * nn.NormActivation                        (eigenn/nn/utils.py:142-150)

Parity unpinned (see oracle/__init__.py): pinned only by first-principles tests.
Everything is plain PyTorch on CPU so that autograd provides the backward.
"""
import collections
import itertools
import math
from fractions import Fraction
from functools import lru_cache
from typing import List, Optional, Tuple

import torch

# --------------------------------------------------------------------------------------
# Irrep / Irreps
# --------------------------------------------------------------------------------------


class Irrep(tuple):
    """(l, p) with p = +1 ('e') or -1 ('o').  Tuple ordering => sort by l, then odd<even."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                p = {"e": 1, "o": -1}[s[-1]]
                l = int(s[:-1])
            elif isinstance(l, tuple):
                l, p = l
        if not (isinstance(l, int) and l >= 0 and p in (1, -1)):
            raise ValueError(f"bad irrep l={l} p={p}")
        return super().__new__(cls, (l, p))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self[0] + 1

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    __str__ = __repr__

    def __mul__(self, other):
        other = Irrep(other)
        p = self.p * other.p
        return [Irrep(l, p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]


class MulIr(tuple):
    def __new__(cls, mul, ir):
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self.mul * self.ir.dim

    def __repr__(self):
        return f"{self.mul}x{self.ir}"


class Irreps(tuple):
    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return super().__new__(cls, irreps)
        out = []
        if isinstance(irreps, Irrep):
            out.append(MulIr(1, irreps))
        elif isinstance(irreps, str):
            if irreps.strip() != "":
                for term in irreps.split("+"):
                    term = term.strip()
                    if "x" in term:
                        mul, ir = term.split("x")
                        out.append(MulIr(int(mul), Irrep(ir.strip())))
                    else:
                        out.append(MulIr(1, Irrep(term)))
        elif irreps is None:
            pass
        else:
            for item in irreps:
                if isinstance(item, (str, Irrep)):
                    out.append(MulIr(1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append(MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p**l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self):
        return sum(mi.mul for mi in self)

    @property
    def ls(self):
        return [mi.ir.l for mi in self for _ in range(mi.mul)]

    @property
    def lmax(self):
        return max(mi.ir.l for mi in self)

    def slices(self):
        s, i = [], 0
        for mi in self:
            s.append(slice(i, i + mi.dim))
            i += mi.dim
        return s

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(mi.ir == ir for mi in self)

    def count(self, ir):
        ir = Irrep(ir)
        return sum(mi.mul for mi in self if mi.ir == ir)

    def __add__(self, other):
        return Irreps(tuple(self) + tuple(Irreps(other)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Irreps(tuple(self)[i])
        return tuple.__getitem__(self, i)

    def __repr__(self):
        return "+".join(f"{mi}" for mi in self)

    __str__ = __repr__

    def simplify(self):
        out = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def sort(self):
        Ret = collections.namedtuple("sort", ["irreps", "p", "inv"])
        out = sorted((ir, i, mul) for i, (mul, ir) in enumerate(self))
        inv = tuple(i for _, i, _ in out)
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return Ret(Irreps([(mul, ir) for ir, _, mul in out]), tuple(p), inv)


# --------------------------------------------------------------------------------------
# Clebsch-Gordan / wigner_3j  (e3nn >= 0.4: o3/_wigner.py)
# --------------------------------------------------------------------------------------


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1 j2 m2 | j3 m3>, Condon-Shortley, via the Racah formula (integer j only)."""
    if m3 != m1 + m2:
        return 0.0
    vmin = max(-j1 + j2 + m3, -j1 + m1, 0)
    vmax = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    f = math.factorial

    C = (
        (2 * j3 + 1)
        * Fraction(
            f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
            f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2),
        )
    ) ** 0.5
    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** (v + j2 + m2) * Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3),
        )
    return float(C * S)


def _real_to_complex(l) -> torch.Tensor:
    q = torch.zeros((2 * l + 1, 2 * l + 1), dtype=torch.complex128)
    s = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _w3j_cached(l1, l2, l3):
    C = torch.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1), dtype=torch.float64)
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                C[l1 + m1, l2 + m2, l3 + m3] = _su2_cg_coeff(l1, m1, l2, m2, l3, m3)
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    Cc = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C.to(torch.complex128))
    assert Cc.imag.abs().max() < 1e-9
    Cr = Cc.real
    return Cr / Cr.norm()


def wigner_3j(l1, l2, l3, dtype=torch.float64) -> torch.Tensor:
    if not abs(l1 - l2) <= l3 <= l1 + l2:
        raise ValueError("triangle inequality violated")
    return _w3j_cached(l1, l2, l3).to(dtype)


# --------------------------------------------------------------------------------------
# Spherical harmonics (o3.SphericalHarmonics(irreps, normalize, normalization))
# --------------------------------------------------------------------------------------


def _sh_unit_norm(lmax: int, x, y, z) -> List[torch.Tensor]:
    """Per-l lists of polynomials with sum_m Y_lm^2 = 1 on the unit sphere (e3nn's
    generated `_spherical_harmonics` before the normalization factor)."""
    out = [[torch.ones_like(x)]]
    if lmax >= 1:
        out.append([x, y, z])
    if lmax >= 2:
        s3 = math.sqrt(3.0)
        y2 = y * y
        x2z2 = x * x + z * z
        sh20 = s3 * x * z
        sh21 = s3 * x * y
        sh22 = y2 - 0.5 * x2z2
        sh23 = s3 * y * z
        sh24 = s3 / 2 * (z * z - x * x)
        out.append([sh20, sh21, sh22, sh23, sh24])
    if lmax >= 3:
        a = math.sqrt(5.0 / 6.0)
        b = math.sqrt(5.0)
        c = math.sqrt(3.0 / 8.0)
        out.append(
            [
                a * (sh20 * z + sh24 * x),
                b * sh20 * y,
                c * (4 * y2 - x2z2) * x,
                0.5 * y * (2 * y2 - 3 * x2z2),
                c * z * (4 * y2 - x2z2),
                b * sh24 * y,
                a * (sh24 * z - sh20 * x),
            ]
        )
    if lmax >= 4:
        raise NotImplementedError("oracle SH restated for l <= 3 only")
    return out


def spherical_harmonics(irreps_sh, vec: torch.Tensor, normalize: bool, normalization: str = "integral"):
    irreps_sh = Irreps(irreps_sh)
    for mul, ir in irreps_sh:
        if ir.p != (-1) ** ir.l:
            raise ValueError("spherical harmonics irreps must have parity (-1)^l")
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    polys = _sh_unit_norm(irreps_sh.lmax, x, y, z)
    cols = []
    for mul, ir in irreps_sh:
        l = ir.l
        if normalization == "component":
            f = math.sqrt(2 * l + 1)
        elif normalization == "integral":
            f = math.sqrt(2 * l + 1) / math.sqrt(4 * math.pi)
        elif normalization == "norm":
            f = 1.0
        else:
            raise ValueError(normalization)
        for _ in range(mul):
            cols += [f * p for p in polys[l]]
    return torch.stack(cols, dim=-1)


# --------------------------------------------------------------------------------------
# normalize2mom (e3nn.math.normalize2mom): Monte-Carlo second-moment constant
# --------------------------------------------------------------------------------------

_N2M_CACHE = {}


def normalize2mom_const(f, key=None) -> float:
    key = key if key is not None else id(f)
    if key not in _N2M_CACHE:
        gen = torch.Generator(device="cpu").manual_seed(0)
        z = torch.randn(1_000_000, generator=gen, dtype=torch.float64)
        cst = f(z).pow(2).mean().pow(-0.5).item()
        _N2M_CACHE[key] = cst
    return _N2M_CACHE[key]


class Normalize2Mom(torch.nn.Module):
    def __init__(self, f, key=None):
        super().__init__()
        self.f = f
        self.cst = normalize2mom_const(f, key)
        self._is_id = abs(self.cst - 1) < 1e-4

    def forward(self, x):
        if self._is_id:
            return self.f(x)
        return self.f(x).mul(self.cst)


# --------------------------------------------------------------------------------------
# TensorProduct
# --------------------------------------------------------------------------------------

Instruction = collections.namedtuple("Instruction", "i_in1 i_in2 i_out mode has_weight path_weight")


class TensorProduct(torch.nn.Module):
    """o3.TensorProduct restated for connection modes 'uvu', 'uvw', 'uuu'.

    irrep_normalization='component', path_normalization='element', all variances 1.
    Weights are consumed in instruction-list order, each block row-major in its
    path_shape ('uvu': [mul1, mul2]; 'uvw': [mul1, mul2, mul_out]).
    """

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, internal_weights=False, shared_weights=False):
        super().__init__()
        self.irreps_in1 = Irreps(irreps_in1)
        self.irreps_in2 = Irreps(irreps_in2)
        self.irreps_out = Irreps(irreps_out)
        ins = []
        for t in instructions:
            t = tuple(t)
            if len(t) == 5:
                t = t + (1.0,)
            ins.append(Instruction(*t))
        self.instructions = ins
        self.shared_weights = shared_weights
        self.internal_weights = internal_weights

        def num_elements(i):
            m1 = self.irreps_in1[i.i_in1].mul
            m2 = self.irreps_in2[i.i_in2].mul
            return {"uvw": m1 * m2, "uvu": m2, "uvv": m1, "uuw": m1, "uuu": 1, "uvuv": 1}[i.mode]

        self.coeffs = []
        self.w_slices = []
        off = 0
        for i in ins:
            mul1, ir1 = self.irreps_in1[i.i_in1]
            mul2, ir2 = self.irreps_in2[i.i_in2]
            mulo, iro = self.irreps_out[i.i_out]
            assert iro in ir1 * ir2, f"{ir1} x {ir2} does not contain {iro}"
            if i.mode == "uvu":
                assert mul1 == mulo
            if i.mode == "uuu":
                assert mul1 == mul2 == mulo
            alpha = iro.dim
            x = sum(num_elements(j) for j in ins if j.i_out == i.i_out)
            if x > 0:
                alpha /= x
            alpha *= i.path_weight
            self.coeffs.append(math.sqrt(alpha))
            if i.has_weight:
                shape = {"uvu": (mul1, mul2), "uvw": (mul1, mul2, mulo), "uuu": (mul1,)}[i.mode]
                n = int(math.prod(shape))
                self.w_slices.append((slice(off, off + n), shape))
                off += n
            else:
                self.w_slices.append(None)
        self.weight_numel = off
        if internal_weights:
            assert shared_weights
            self.weight = torch.nn.Parameter(torch.randn(off))
        else:
            self.weight = None

    def forward(self, x1, x2, weight: Optional[torch.Tensor] = None):
        if weight is None:
            weight = self.weight
        z = x1.shape[0]
        sl1, sl2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        outs = [None] * len(self.irreps_out)
        for i, c, ws in zip(self.instructions, self.coeffs, self.w_slices):
            mul1, ir1 = self.irreps_in1[i.i_in1]
            mul2, ir2 = self.irreps_in2[i.i_in2]
            mulo, iro = self.irreps_out[i.i_out]
            a = x1[:, sl1[i.i_in1]].reshape(z, mul1, ir1.dim)
            b = x2[:, sl2[i.i_in2]].reshape(z, mul2, ir2.dim)
            C = wigner_3j(ir1.l, ir2.l, iro.l, dtype=x1.dtype).to(x1.device)
            if ws is not None:
                wsl, shape = ws
                if self.shared_weights:
                    w = weight[wsl].reshape(shape)
                else:
                    w = weight[:, wsl].reshape((z,) + shape)
            if i.mode == "uvu":
                xx = torch.einsum("zui,zvj->zuvij", a, b)
                if self.shared_weights:
                    r = torch.einsum("uv,ijk,zuvij->zuk", w, C, xx)
                else:
                    r = torch.einsum("zuv,ijk,zuvij->zuk", w, C, xx)
            elif i.mode == "uvw":
                assert self.shared_weights
                xx = torch.einsum("zui,zvj->zuvij", a, b)
                r = torch.einsum("uvw,ijk,zuvij->zwk", w, C, xx)
            elif i.mode == "uuu":
                r = torch.einsum("ijk,zui,zuj->zuk", C, a, b)
                if ws is not None:
                    r = r * (w if self.shared_weights else w)[..., None]
            else:
                raise NotImplementedError(i.mode)
            r = (c * r).reshape(z, mulo * iro.dim)
            outs[i.i_out] = r if outs[i.i_out] is None else outs[i.i_out] + r
        cols = []
        for k, (mul, ir) in enumerate(self.irreps_out):
            cols.append(outs[k] if outs[k] is not None else x1.new_zeros(z, mul * ir.dim))
        return torch.cat(cols, dim=-1) if cols else x1.new_zeros(z, 0)


class FullyConnectedTensorProduct(TensorProduct):
    def __init__(self, irreps_in1, irreps_in2, irreps_out):
        i1 = Irreps(irreps_in1).simplify()
        i2 = Irreps(irreps_in2).simplify()
        io = Irreps(irreps_out).simplify()
        instr = [
            (a, b, c, "uvw", True, 1.0)
            for a, (_, ir1) in enumerate(i1)
            for b, (_, ir2) in enumerate(i2)
            for c, (_, iro) in enumerate(io)
            if iro in ir1 * ir2
        ]
        super().__init__(i1, i2, io, instr, internal_weights=True, shared_weights=True)


class Linear(torch.nn.Module):
    """o3.Linear: all (i_in, i_out) pairs with equal irrep; weight blocks [mul_in, mul_out]
    flattened in instruction order; path weight 1/sqrt(sum of mul_in reaching i_out)."""

    def __init__(self, irreps_in, irreps_out):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_out)
        self.instructions = [
            (a, b)
            for a, (_, ir_in) in enumerate(self.irreps_in)
            for b, (_, ir_out) in enumerate(self.irreps_out)
            if ir_in == ir_out
        ]
        self.pw = []
        off = 0
        self.w_slices = []
        for a, b in self.instructions:
            fan = sum(self.irreps_in[a2].mul for a2, b2 in self.instructions if b2 == b)
            self.pw.append(fan ** (-0.5))
            n = self.irreps_in[a].mul * self.irreps_out[b].mul
            self.w_slices.append(slice(off, off + n))
            off += n
        self.weight_numel = off
        self.weight = torch.nn.Parameter(torch.randn(off))

    def forward(self, x):
        z = x.shape[0]
        sl = self.irreps_in.slices()
        outs = [None] * len(self.irreps_out)
        for (a, b), pw, ws in zip(self.instructions, self.pw, self.w_slices):
            mul_in, ir = self.irreps_in[a]
            mul_out, _ = self.irreps_out[b]
            w = self.weight[ws].reshape(mul_in, mul_out)
            xb = x[:, sl[a]].reshape(z, mul_in, ir.dim)
            r = pw * torch.einsum("uw,zui->zwi", w, xb)
            r = r.reshape(z, mul_out * ir.dim)
            outs[b] = r if outs[b] is None else outs[b] + r
        cols = []
        for k, (mul, ir) in enumerate(self.irreps_out):
            cols.append(outs[k] if outs[k] is not None else x.new_zeros(z, mul * ir.dim))
        return torch.cat(cols, dim=-1)


# --------------------------------------------------------------------------------------
# FullyConnectedNet
# --------------------------------------------------------------------------------------


class _FCLayer(torch.nn.Module):
    def __init__(self, h_in, h_out, act):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(h_in, h_out))
        self.act = act
        self.h_in = h_in

    def forward(self, x):
        w = self.weight / (self.h_in**0.5)
        x = x @ w
        if self.act is not None:
            x = self.act(x)
        return x


class FullyConnectedNet(torch.nn.Sequential):
    def __init__(self, hs, act=None, act_key=None):
        super().__init__()
        self.hs = list(hs)
        if act is not None:
            act = Normalize2Mom(act, act_key)
        for i, (h1, h2) in enumerate(zip(self.hs, self.hs[1:])):
            last = i == len(self.hs) - 2
            setattr(self, f"layer{i}", _FCLayer(h1, h2, None if last else act))


# --------------------------------------------------------------------------------------
# Gate
# --------------------------------------------------------------------------------------


def _act_parity(act, p_in):
    """e3nn.nn.Activation parity rule: even act -> 0e; odd act keeps parity."""
    if p_in == 1:
        return 1
    x = torch.linspace(0, 10, 256, dtype=torch.float64)
    a1, a2 = act(x), act(-x)
    if (a1 - a2).abs().max() < 1e-5:
        return 1
    if (a1 + a2).abs().max() < 1e-5:
        return -1
    raise ValueError("activation of an odd scalar must be even or odd")


class Activation(torch.nn.Module):
    def __init__(self, irreps_in, acts, act_keys=None):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        assert len(self.irreps_in) == len(acts)
        act_keys = act_keys or [None] * len(acts)
        self.acts = torch.nn.ModuleList()
        out = []
        for (mul, ir), a, k in zip(self.irreps_in, acts, act_keys):
            if a is None:
                self.acts.append(torch.nn.Identity())
                out.append((mul, ir))
                continue
            if ir.l != 0:
                raise ValueError("Activation: cannot apply an activation to a non-scalar")
            n = Normalize2Mom(a, k)
            self.acts.append(n)
            out.append((mul, (0, _act_parity(n, ir.p))))
        self.irreps_out = Irreps(out)

    def forward(self, x):
        cols = []
        for sl, a in zip(self.irreps_in.slices(), self.acts):
            cols.append(a(x[:, sl]))
        return torch.cat(cols, dim=-1) if cols else x


class Gate(torch.nn.Module):
    def __init__(self, irreps_scalars, act_scalars, irreps_gates, act_gates, irreps_gated, keys_scalars=None, keys_gates=None):
        super().__init__()
        self.irreps_scalars = Irreps(irreps_scalars)
        self.irreps_gates = Irreps(irreps_gates)
        self.irreps_gated = Irreps(irreps_gated)
        if len(self.irreps_gates) > 0 and self.irreps_gates.lmax > 0:
            raise ValueError("gates must be scalars")
        if len(self.irreps_scalars) > 0 and self.irreps_scalars.lmax > 0:
            raise ValueError("scalars must be scalars")
        if self.irreps_gates.num_irreps != self.irreps_gated.num_irreps:
            raise ValueError("number of gates != number of gated irreps")
        self.irreps_in = (self.irreps_scalars + self.irreps_gates + self.irreps_gated).simplify()
        self.act_scalars = Activation(self.irreps_scalars, act_scalars, keys_scalars)
        self.act_gates = Activation(self.irreps_gates, act_gates, keys_gates)
        # ElementwiseTensorProduct(irreps_gated, act_gates.irreps_out): plain multiply,
        # output parity = gated parity * gate parity
        gate_p = []
        for mul, ir in self.act_gates.irreps_out:
            gate_p += [ir.p] * mul
        out = []
        u = 0
        for mul, ir in self.irreps_gated.simplify():
            # split on gate-parity changes, as ElementwiseTensorProduct aligns blocks
            start = 0
            while start < mul:
                p = gate_p[u + start]
                end = start
                while end < mul and gate_p[u + end] == p:
                    end += 1
                out.append((end - start, (ir.l, ir.p * p)))
                start = end
            u += mul
        self.irreps_out = self.act_scalars.irreps_out + Irreps(out)

    def forward(self, x):
        ns, ng = self.irreps_scalars.dim, self.irreps_gates.dim
        scalars, gates, gated = x[:, :ns], x[:, ns : ns + ng], x[:, ns + ng :]
        scalars = self.act_scalars(scalars)
        if ng == 0:
            return scalars
        gates = self.act_gates(gates)
        cols = [scalars]
        u = 0
        for (mul, ir), sl in zip(self.irreps_gated, self.irreps_gated.slices()):
            blk = gated[:, sl].reshape(-1, mul, ir.dim) * gates[:, u : u + mul, None]
            cols.append(blk.reshape(-1, mul * ir.dim))
            u += mul
        return torch.cat(cols, dim=-1)


# --------------------------------------------------------------------------------------
# Wigner-D of the real basis from the SH themselves (test helper)
# --------------------------------------------------------------------------------------


def wigner_D_from_sh(l: int, R: torch.Tensor) -> torch.Tensor:
    """D^l(R) with Y_l(R v) = D^l(R) Y_l(v), obtained by least squares from the SH
    polynomials, so that it lives in exactly the basis the SH/CG use."""
    g = torch.Generator().manual_seed(1234 + l)
    v = torch.randn(64 + 8 * l, 3, generator=g, dtype=torch.float64)
    ir = Irreps([(1, (l, (-1) ** l))])
    A = spherical_harmonics(ir, v, True, "component")
    B = spherical_harmonics(ir, v @ R.T, True, "component")
    # B = A @ D^T
    Dt = torch.linalg.lstsq(A, B).solution
    return Dt.T


def irreps_D(irreps, R: torch.Tensor, parity_of_R: int = 1) -> torch.Tensor:
    """Block-diagonal representation matrix of `irreps` for a proper rotation R times an
    optional inversion (parity_of_R = -1)."""
    irreps = Irreps(irreps)
    blocks = []
    for mul, ir in irreps:
        D = wigner_D_from_sh(ir.l, R)
        if parity_of_R == -1 and ir.p == -1:
            D = -D
        blocks.append(torch.kron(torch.eye(mul, dtype=torch.float64), D))
    return torch.block_diag(*blocks) if blocks else torch.zeros(0, 0, dtype=torch.float64)


class NormActivation(torch.nn.Module):
    """e3nn.nn.NormActivation(irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False),
    restated in e3nn's own op order (recollection of e3nn 0.4.x `e3nn/nn/_normact.py`; not under
    /root/reference, PARITY UNPINNED like the rest of this file):
        norms = o3.Norm(irreps_in, squared=(epsilon is not None))(x)        # per irrep copy
        if epsilon: norms[norms < eps^2] = eps^2 (in place: no gradient through clamped entries); norms = sqrt(norms)
        scalings = scalar_nonlinearity(norms [+ biases]);  if normalize: scalings = scalings / norms
        out = o3.ElementwiseTensorProduct(norm irreps (0e per copy), irreps_in)(scalings, x)
    The elementwise product 0e x l -> l with 'component' normalisation is a plain multiply
    (sqrt(2l+1) * wigner_3j(0, l, l) = identity).  The non-linearity is applied as is (NormActivation
    does not wrap it in normalize2mom).  bias=False only (what the reference passes)."""

    def __init__(self, irreps_in, scalar_nonlinearity, normalize=True, epsilon=None, bias=False):
        super().__init__()
        self.irreps_in = Irreps(irreps_in)
        self.irreps_out = Irreps(irreps_in)
        if epsilon is None and normalize:
            epsilon = 1e-8
        elif epsilon is not None and not normalize:
            raise ValueError("epsilon and normalize = False don't make sense together")
        if bias:
            raise NotImplementedError("oracle NormActivation: bias=True not restated")
        self._eps_squared = 0.0 if epsilon is None else epsilon * epsilon
        self.scalar_nonlinearity = scalar_nonlinearity
        self.normalize = normalize

    def forward(self, x):
        cols = []
        for (mul, ir), sl in zip(self.irreps_in, self.irreps_in.slices()):
            if mul == 0:
                continue
            blk = x[:, sl].reshape(-1, mul, ir.dim)
            norms = blk.pow(2).sum(-1) if self._eps_squared > 0 else blk.pow(2).sum(-1).sqrt()
            if self._eps_squared > 0:
                # masked in-place assignment: a constant where clamped (zero gradient), the value elsewhere
                norms = torch.where(norms < self._eps_squared, torch.full_like(norms, self._eps_squared), norms)
                norms = norms.sqrt()
            scalings = self.scalar_nonlinearity(norms)
            if self.normalize:
                scalings = scalings / norms
            cols.append((blk * scalings[:, :, None]).reshape(x.shape[0], -1))
        return torch.cat(cols, dim=-1) if cols else x
