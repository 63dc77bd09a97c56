"""Real Clebsch-Gordan tables for the 'uvu' tensor product (host side, numpy, fp64).

The reference obtains these from `e3nn.o3.wigner_3j` inside `o3.TensorProduct`
(eigenn/nn/utils.py:231-238).  The tables are built once per plan on the host and
uploaded to the device; the kernels only ever see fp32 copies.

Convention (e3nn >= 0.4): complex Condon-Shortley coefficients <l1 m1 l2 m2|l3 m3>,
rotated to the real basis in which the spherical harmonics of `csrc/tfc_edge.hip`
live (y is the polar axis; l=1 is ordered x, y, z), then scaled to unit Frobenius
norm.
"""
from __future__ import annotations

from fractions import Fraction
from functools import lru_cache
from math import factorial as fact

import numpy as np


def _cs_coefficient(j1: int, m1: int, j2: int, m2: int, j: int, m: int) -> float:
    """Condon-Shortley <j1 m1 j2 m2 | j m> through the Wigner 3j closed form."""
    if m1 + m2 != m or not (abs(j1 - j2) <= j <= j1 + j2):
        return 0.0
    if abs(m1) > j1 or abs(m2) > j2 or abs(m) > j:
        return 0.0
    # 3j symbol (j1 j2 j; m1 m2 -m) by Racah's formula, exact rational under the root
    tri = Fraction(fact(j1 + j2 - j) * fact(j1 - j2 + j) * fact(-j1 + j2 + j), fact(j1 + j2 + j + 1))
    pref = tri * fact(j1 + m1) * fact(j1 - m1) * fact(j2 + m2) * fact(j2 - m2) * fact(j + m) * fact(j - m)
    tot = Fraction(0)
    for t in range(0, j1 + j2 + j + 1):
        d = [t, j - j2 + t + m1, j - j1 + t - m2, j1 + j2 - j - t, j1 - t - m1, j2 - t + m2]
        if min(d) < 0:
            continue
        den = 1
        for q in d:
            den *= fact(q)
        tot += Fraction((-1) ** t, den)
    threej = (-1) ** (j1 - j2 + m) * float(tot) * float(pref) ** 0.5
    return (-1) ** (j1 - j2 + m) * (2 * j + 1) ** 0.5 * threej


def _u_real_to_complex(l: int) -> np.ndarray:
    u = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    r = 2**-0.5
    for m in range(1, l + 1):
        u[l - m, l + m] = r
        u[l - m, l - m] = -1j * r
        u[l + m, l + m] = (-1) ** m * r
        u[l + m, l - m] = 1j * (-1) ** m * r
    u[l, l] = 1.0
    return (-1j) ** l * u


@lru_cache(maxsize=None)
def real_cg(l1: int, l2: int, l3: int) -> np.ndarray:
    """C[m1, m2, m3], shape (2l1+1, 2l2+1, 2l3+1), float64, unit Frobenius norm."""
    if not abs(l1 - l2) <= l3 <= l1 + l2:
        raise ValueError(f"({l1},{l2},{l3}) violates the triangle rule")
    cs = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for a in range(-l1, l1 + 1):
        for b in range(-l2, l2 + 1):
            if abs(a + b) <= l3:
                cs[l1 + a, l2 + b, l3 + a + b] = _cs_coefficient(l1, a, l2, b, l3, a + b)
    u1, u2, u3 = _u_real_to_complex(l1), _u_real_to_complex(l2), _u_real_to_complex(l3)
    c = np.einsum("ia,jb,kc,ijk->abc", u1, u2, np.conj(u3), cs.astype(np.complex128))
    if np.abs(c.imag).max() > 1e-9:
        raise AssertionError("real CG has an imaginary part")
    c = np.ascontiguousarray(c.real)
    c[np.abs(c) < 1e-14] = 0.0
    c /= np.linalg.norm(c)
    c.setflags(write=False)
    return c
