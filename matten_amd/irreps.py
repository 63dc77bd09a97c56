"""Irreps bookkeeping for the tensor-field-convolution path (host side, no e3nn).

The reference gets this algebra from `e3nn.o3.Irreps` (imported at
eigenn/nn/point_conv.py:4, eigenn/nn/utils.py:6, eigenn/data/irreps.py:10).  Only the
operations the hot path's constructors use are provided: parsing, `dim`, `ls`,
`in`, `+`, `==`, `ir * ir`, `simplify()`, `sort()`, `slices()`.

An irrep is the pair (l, p), p = +1 for even ('e'), -1 for odd ('o').  A feature
tensor with irreps `sum_b mul_b x (l_b, p_b)` stores, along its last axis, the
blocks b one after the other, each block row-major as [mul_b, 2 l_b + 1].
"""
from __future__ import annotations

from typing import Iterable, List, NamedTuple, Tuple


class Irrep(NamedTuple):
    l: int
    p: int

    @staticmethod
    def of(x) -> "Irrep":
        if isinstance(x, Irrep):
            return x
        if isinstance(x, tuple) and len(x) == 2:
            l, p = int(x[0]), int(x[1])
        else:
            s = str(x).strip()
            if len(s) < 2 or s[-1] not in "eo" or not s[:-1].isdigit():
                raise ValueError(f"cannot parse irrep {x!r}")
            l, p = int(s[:-1]), (1 if s[-1] == "e" else -1)
        if l < 0 or p not in (1, -1):
            raise ValueError(f"invalid irrep l={l} p={p}")
        return Irrep(l, p)

    @property
    def dim(self) -> int:
        return 2 * self.l + 1

    def __str__(self) -> str:
        return f"{self.l}{'e' if self.p > 0 else 'o'}"

    __repr__ = __str__

    def __mul__(self, other) -> List["Irrep"]:  # selection rule, ascending l
        o = Irrep.of(other)
        return [Irrep(l, self.p * o.p) for l in range(abs(self.l - o.l), self.l + o.l + 1)]


class MulIrrep(NamedTuple):
    mul: int
    ir: Irrep

    @property
    def dim(self) -> int:
        return self.mul * self.ir.dim

    def __str__(self) -> str:
        return f"{self.mul}x{self.ir}"

    __repr__ = __str__


class Irreps(tuple):
    """Immutable ordered list of (mul, Irrep)."""

    def __new__(cls, spec=None):
        if isinstance(spec, Irreps):
            return spec
        items: List[MulIrrep] = []
        if spec is None:
            pass
        elif isinstance(spec, Irrep):
            items.append(MulIrrep(1, spec))
        elif isinstance(spec, (list, tuple)):
            for it in spec:
                if isinstance(it, (str, Irrep)):
                    items.append(MulIrrep(1, Irrep.of(it)))
                else:
                    mul, ir = it
                    if int(mul) < 0:
                        raise ValueError("negative multiplicity")
                    items.append(MulIrrep(int(mul), Irrep.of(ir)))
        else:
            # str, or any object (e.g. a real e3nn Irreps) whose str() is an irreps string
            text = str(spec).strip()
            if text:
                for term in text.split("+"):
                    term = term.strip()
                    if "x" in term:
                        mul, ir = term.split("x", 1)
                        items.append(MulIrrep(int(mul), Irrep.of(ir)))
                    else:
                        items.append(MulIrrep(1, Irrep.of(term)))
        if any(m.mul < 0 for m in items):
            raise ValueError(f"negative multiplicity in {spec!r}")
        return super().__new__(cls, items)

    @staticmethod
    def spherical_harmonics(lmax: int, p: int = -1) -> "Irreps":
        return Irreps([(1, (l, p**l)) for l in range(lmax + 1)])

    # ---- sizes
    @property
    def dim(self) -> int:
        return sum(m.dim for m in self)

    @property
    def num_irreps(self) -> int:
        return sum(m.mul for m in self)

    @property
    def ls(self) -> List[int]:
        return [m.ir.l for m in self for _ in range(m.mul)]

    @property
    def lmax(self) -> int:
        if not len(self):
            raise ValueError("lmax of empty irreps")
        return max(m.ir.l for m in self)

    def offsets(self) -> List[int]:
        out, o = [], 0
        for m in self:
            out.append(o)
            o += m.dim
        return out

    def slices(self) -> List[slice]:
        return [slice(o, o + m.dim) for o, m in zip(self.offsets(), self)]

    # ---- algebra
    def __contains__(self, ir) -> bool:
        try:
            ir = Irrep.of(ir)
        except ValueError:
            return False
        return any(m.ir == ir for m in self)

    def __add__(self, other) -> "Irreps":
        return Irreps(list(self) + list(Irreps(other)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Irreps(list(self)[i])
        return tuple.__getitem__(self, i)

    def __eq__(self, other) -> bool:
        if other is None:
            return False
        try:
            return tuple(self) == tuple(Irreps(other))
        except (ValueError, TypeError):
            return False

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    __hash__ = tuple.__hash__

    def __str__(self) -> str:
        return "+".join(str(m) for m in self)

    __repr__ = __str__

    def simplify(self) -> "Irreps":
        """Merge ADJACENT blocks of equal irrep; drop empty blocks.  No reordering."""
        out: List[Tuple[int, Irrep]] = []
        for mul, ir in self:
            if out and out[-1][1] == ir:
                out[-1] = (out[-1][0] + mul, ir)
            elif mul > 0:
                out.append((mul, ir))
        return Irreps(out)

    def sort(self):
        """Stable sort by (l, p) - odd before even.  Returns (irreps, p, inv) with
        p[old] = new and inv[new] = old, as e3nn's `Irreps.sort()`."""
        order = sorted(range(len(self)), key=lambda i: (self[i].ir, i))
        p = [0] * len(order)
        for new, old in enumerate(order):
            p[old] = new
        return Irreps([self[i] for i in order]), tuple(p), tuple(order)


def tp_path_exists(irreps_in1, irreps_in2, ir_out) -> bool:
    """eigenn/nn/utils.py:360-369"""
    a, b, o = Irreps(irreps_in1).simplify(), Irreps(irreps_in2).simplify(), Irrep.of(ir_out)
    return any(o in (m1.ir * m2.ir) for m1 in a for m2 in b)
