"""CPU: first-principles pins of the oracle (the reference holds no golden vector for this
path, SURVEY.md §4 - "parity unpinned" - so these are what defends the restatement):
known Clebsch-Gordan / spherical-harmonic / normalize2mom answers, the
structural irreps tables of App. B, rotation + inversion equivariance of the whole layer stack
in fp64, and closed-form checks of the radial basis."""
import itertools
import math

import pytest
import torch

from oracle import e3ref as e3
from oracle import eigenn_ref as R


def random_rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


# ------------------------------------------------------------------------------ CG
def test_cg_known_answers():
    s = math.sqrt
    C = e3.wigner_3j
    assert C(0, 0, 0).item() == pytest.approx(1.0)
    assert torch.allclose(C(1, 1, 0)[:, :, 0], torch.eye(3, dtype=torch.float64) / s(3))
    assert torch.allclose(C(0, 1, 1)[0], torch.eye(3, dtype=torch.float64) / s(3))
    for l in range(4):
        assert torch.allclose(C(l, 0, l)[:, 0, :], torch.eye(2 * l + 1, dtype=torch.float64) / s(2 * l + 1))
    assert C(1, 1, 1)[0, 1, 2].item() == pytest.approx(1 / s(6))
    assert C(1, 1, 2)[0, 2, 0].item() == pytest.approx(1 / s(10))
    assert C(1, 1, 2)[1, 1, 2].item() == pytest.approx(2 / s(30))
    assert C(2, 1, 1)[2, 1, 1].item() == pytest.approx(2 / s(30))
    assert C(1, 2, 3)[1, 2, 3].item() == pytest.approx(0.292770, abs=1e-6)


def test_su2_cg_against_sympy():
    from sympy import S
    from sympy.physics.quantum.cg import CG

    for j1, j2, j3 in [(1, 1, 2), (2, 1, 1), (2, 2, 2), (3, 2, 1), (3, 3, 0), (2, 3, 3), (3, 3, 3)]:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                m3 = m1 + m2
                if abs(m3) > j3:
                    continue
                ref = float(CG(S(j1), S(m1), S(j2), S(m2), S(j3), S(m3)).doit())
                assert e3._su2_cg_coeff(j1, m1, j2, m2, j3, m3) == pytest.approx(ref, abs=1e-12)


@pytest.mark.parametrize("l1,l2", list(itertools.product(range(4), repeat=2)))
def test_cg_invariant_and_product_rule(l1, l2):
    """C = (D1 x D2 x D3) C in the basis of the SH, and SH ~ +C.SH.SH (e3nn's sign convention)."""
    Rm = random_rotation(l1 * 7 + l2)
    v = torch.randn(200, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
        C = e3.wigner_3j(l1, l2, l3)
        D1, D2, D3 = (e3.wigner_D_from_sh(l, Rm) for l in (l1, l2, l3))
        C2 = torch.einsum("ia,jb,kc,abc->ijk", D1, D2, D3, C)
        assert (C2 - C).abs().max() < 1e-11
        assert C.norm().item() == pytest.approx(1.0)
        if (l1 + l2 + l3) % 2 == 0:
            Y = [e3.spherical_harmonics([(1, (l, (-1) ** l))], v, True, "component") for l in (l1, l2, l3)]
            prod = torch.einsum("ijk,zi,zj->zk", C, Y[0], Y[1])
            assert (prod * Y[2]).sum() > 0


# ------------------------------------------------------------------------------ SH
def test_sh_norm_parity_axis():
    v = torch.randn(500, 3, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    Y = e3.spherical_harmonics("0e+1o+2e+3o", v, True, "component")
    sl = [slice(0, 1), slice(1, 4), slice(4, 9), slice(9, 16)]
    for l, s in enumerate(sl):
        assert torch.allclose((Y[:, s] ** 2).sum(1), torch.full((500,), 2.0 * l + 1, dtype=torch.float64))
    Ym = e3.spherical_harmonics("0e+1o+2e+3o", -v, True, "component")
    for l, s in enumerate(sl):
        assert torch.allclose(Ym[:, s], (-1) ** l * Y[:, s])
    ey = e3.spherical_harmonics("0e+1o+2e+3o", torch.tensor([[0.0, 2.0, 0.0]], dtype=torch.float64), True, "component")[0]
    for l, s in enumerate(sl):
        blk = ey[s]
        assert blk[l].item() == pytest.approx(math.sqrt(2 * l + 1))
        blk = blk.clone()
        blk[l] = 0
        assert blk.abs().max() < 1e-12
    # D^1(R) = R : l=1 is ordered x, y, z
    Rm = random_rotation(5)
    assert torch.allclose(e3.wigner_D_from_sh(1, Rm), Rm, atol=1e-12)
    # zero vector -> zeros beyond l=0 (F.normalize)
    z = e3.spherical_harmonics("0e+1o+2e", torch.zeros(1, 3, dtype=torch.float64), True, "component")
    assert z[0, 0] == 1 and z[0, 1:].abs().max() == 0


def test_normalize2mom_constants():
    import matten_amd.nn.e3 as pe3

    fns = {"ssp": R.ssp, "silu": torch.nn.functional.silu, "tanh": torch.tanh, "abs": torch.abs, "sigmoid": torch.sigmoid}
    expect = {"ssp": 1.878204668541552, "silu": 1.6791767923989418, "tanh": 1.5937334472592692, "abs": 1.001110600838467,
              "sigmoid": 1.8467055342154763}
    for k, f in fns.items():
        c = e3.normalize2mom_const(f, key="t_" + k)
        assert c == pytest.approx(expect[k], rel=1e-12)
        assert pe3.NORMALIZE2MOM[k] == pytest.approx(c, rel=1e-12)  # the baked product constants


# ------------------------------------------------------------------------------ structure (App. B)
def build_layers(emb, sh, conv, n_layers):
    irr = {R.NODE_FEATURES: f"{emb}x0e", R.NODE_ATTRS: f"{emb}x0e", R.EDGE_EMBEDDING: "8x0e", R.EDGE_ATTRS: sh, R.EDGE_INDEX: None}
    out = []
    for _ in range(n_layers):
        L = R.PointConvMessagePassing(irr, conv, message_kwargs=dict(fc_num_hidden_layers=2, fc_hidden_size=64),
                                      update_kwargs=dict(use_self_connection=True, avg_num_neighbors=10.0))
        out.append(L)
        irr = L.irreps_out
    return out


def test_minimal_yaml_irreps_flow():
    Ls = build_layers(32, "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", 3)
    exp = [
        ("56x0e+16x1o+8x2e", 3, 96, 288, "32x0e+32x1o+32x2e", "32x0e+16x1o+8x2e", (1024, 2560, 57344)),
        ("80x0e+16x1o+16x1e+8x2o+8x2e", 15, 240, 784, "56x0e+72x1o+24x1e+24x2o+64x2e", "32x0e+16x1o+16x1e+8x2o+8x2e", (1344, 6720, 92160)),
        ("32x0o+80x0e+16x1o+16x1e+8x2o+8x2e", 27, 384, 1280, "24x0o+56x0e+96x1o+64x1e+56x2o+88x2e",
         "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", (1664, 8960, 102400)),
    ]
    for L, (before, paths, W, dmid, tpout, after, npar) in zip(Ls, exp):
        tp = L.message_fn.tp
        assert str(L.equivariant_nonlinear.irreps_in) == before
        assert len(tp.tp.instructions) == paths and tp.tp.weight_numel == W and tp.irreps_mid.dim == dmid
        assert str(tp.irreps_out) == tpout and str(L.irreps_out[R.NODE_FEATURES]) == after
        assert (L.message_fn.linear_1.weight.numel(), L.update_fn.linear_2.weight.numel(), L.update_fn.self_connection.weight.numel()) == npar
        assert not L.use_resnet
        assert str(L.irreps_out[R.EDGE_MESSAGE]) == tpout
    ins = [(i.i_in1, i.i_in2, i.i_out) for i in Ls[1].message_fn.tp.tp.instructions]
    assert ins == [(0, 0, 0), (0, 1, 3), (0, 2, 11), (1, 0, 4), (1, 1, 1), (1, 1, 7), (1, 1, 12), (1, 2, 5), (1, 2, 9), (2, 0, 13),
                   (2, 1, 6), (2, 1, 10), (2, 2, 2), (2, 2, 8), (2, 2, 14)]


def test_config3_and_4_shapes():
    Ls = build_layers(64, "0e+1o+2e", "64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", 3)
    got = [(len(L.message_fn.tp.tp.instructions), L.message_fn.tp.tp.weight_numel, L.irreps_in[R.NODE_FEATURES].dim,
            L.message_fn.tp.irreps_mid.dim, L.equivariant_nonlinear.irreps_in.dim) for L in Ls]
    assert got == [(3, 192, 64, 576, 704), (15, 960, 576, 3264, 1344), (27, 1728, 1088, 5952, 1408)]
    assert Ls[2].irreps_out[R.NODE_FEATURES].dim == 1152
    Ls = build_layers(128, "0e+1o+2e+3o", "128x0o+128x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e", 3)
    got = [(len(L.message_fn.tp.tp.instructions), L.message_fn.tp.tp.weight_numel, L.irreps_in[R.NODE_FEATURES].dim,
            L.message_fn.tp.irreps_mid.dim, L.equivariant_nonlinear.irreps_in.dim) for L in Ls]
    assert got == [(4, 512, 128, 2048, 2432), (34, 4352, 2048, 19968, 4736), (64, 8192, 3968, 37888, 4864)]


def test_constructor_errors():
    with pytest.raises(ValueError):
        R.PointConvMessage({R.NODE_FEATURES: "4x0e"}, {R.NODE_FEATURES: "4x0e"})  # missing keys
    with pytest.raises(ValueError):
        R.PointConvMessage({R.NODE_FEATURES: "4x0e", R.EDGE_EMBEDDING: "4x1o", R.EDGE_ATTRS: "0e", R.EDGE_INDEX: None},
                           {R.NODE_FEATURES: "4x0e"})  # embedding must be 0e
    with pytest.raises(AssertionError):
        R.UVUTensorProduct("4x0e", "1o", "4x0e", mlp_input_size=4)  # no path
    # reference's own constructor smoke test (tests/nn/test_convolution.py:8-21)
    R.PointConv(irreps_in={R.NODE_FEATURES: "4x1e+2x0e", R.NODE_ATTRS: "4x1e+2x0e", R.EDGE_ATTRS: "4x1e+2x0e", R.EDGE_EMBEDDING: "4x0e"},
                irreps_out={R.NODE_FEATURES: "8x1e+4x0e"})


# ------------------------------------------------------------------------------ equivariance
@pytest.mark.parametrize("parity", [1, -1])
def test_stack_equivariance_fp64(parity):
    from tests.parity import CONFIGS, Stack, _prep, make_batch
    from matten_amd import keys as K

    cfg = dict(CONFIGS["lmax3"], layers=2)
    torch.manual_seed(1)
    model = Stack("oracle", cfg).double()
    batch = make_batch(cfg)
    Rm = random_rotation(11) * parity
    b1 = _prep(batch, "cpu", torch.float64)
    f1, e1, _ = model(b1, cfg["graphs"])
    b2 = _prep(batch, "cpu", torch.float64)
    b2[K.POSITIONS] = (b2[K.POSITIONS].detach() @ Rm.T).requires_grad_(True)
    b2[K.CELL] = (b2[K.CELL].view(-1, 3, 3) @ Rm.T).reshape(-1, 3)
    f2, e2, _ = model(b2, cfg["graphs"])
    irreps_out = model.convs[-1].irreps_out[K.NODE_FEATURES]
    D = e3.irreps_D(irreps_out, Rm * parity, parity)
    assert (f2 - f1 @ D.T).abs().max() < 1e-9 * max(1.0, f1.abs().max().item())


def test_forces_rotate_with_invariant_readout():
    from tests.parity import CONFIGS, Stack, _prep, make_batch
    from matten_amd import keys as K

    cfg = dict(CONFIGS["smoke"], layers=2)
    torch.manual_seed(2)
    model = Stack("oracle", cfg).double()
    irreps_out = model.convs[-1].irreps_out[K.NODE_FEATURES]
    # invariant read-out: keep even scalars only
    mask = torch.zeros(irreps_out.dim, dtype=torch.float64)
    off = 0
    for mul, ir in irreps_out:
        if ir.l == 0 and ir.p == 1:
            mask[off : off + mul] = 1
        off += mul * ir.dim
    model.readout = model.readout * mask
    batch = make_batch(cfg)
    Rm = random_rotation(13)
    outs = []
    for rot in (torch.eye(3, dtype=torch.float64), Rm, -Rm):
        b = _prep(batch, "cpu", torch.float64)
        b[K.POSITIONS] = (b[K.POSITIONS].detach() @ rot.T).requires_grad_(True)
        b[K.CELL] = (b[K.CELL].view(-1, 3, 3) @ rot.T).reshape(-1, 3)
        _, e, _ = model(b, cfg["graphs"])
        (g,) = torch.autograd.grad(e.sum(), [b[K.POSITIONS]])
        outs.append((e.detach(), g, rot))
    e0, g0, _ = outs[0]
    for e, g, rot in outs[1:]:
        assert (e - e0).abs().max() < 1e-10 * max(1.0, e0.abs().max().item())
        assert (g - g0 @ rot.T).abs().max() < 1e-10 * max(1.0, g0.abs().max().item())


# ------------------------------------------------------------------------------ radial basis
def test_radial_basis_closed_form():
    rb = R.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0}).double()
    r = torch.tensor([0.5, 1.0, 2.5, 4.999, 5.0, 6.0], dtype=torch.float64)
    data = {R.EDGE_VECTORS: torch.stack([r, torch.zeros_like(r), torch.zeros_like(r)], 1)}
    emb = rb(data)[R.EDGE_EMBEDDING]
    u = r / 5.0
    env = (1 - 28 * u**6 + 48 * u**7 - 21 * u**8) * (r < 5.0)
    w = rb.basis.bessel_weights.detach()  # n * pi, as initialised in fp32 by the reference
    assert torch.allclose(w, torch.arange(1, 9, dtype=torch.float64) * math.pi, rtol=1e-6)
    ref = (2 / 5.0) * torch.sin(w[None] * r[:, None] / 5.0) / r[:, None] * env[:, None]
    assert torch.allclose(emb, ref, atol=1e-12)
    assert emb[4:].abs().max() == 0  # strict cutoff


def test_gradcheck_uvu_tensor_product():
    tp = R.UVUTensorProduct("2x0e+2x1o", "0e+1o+2e", "2x0e+2x1o+2x1e+2x2e", mlp_input_size=3, mlp_hidden_size=4).double()
    x = torch.randn(5, 8, dtype=torch.float64, requires_grad=True)
    y = torch.randn(5, 9, dtype=torch.float64, requires_grad=True)
    e = torch.randn(5, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b, c: tp(a, b, c), (x, y, e), atol=1e-8)


def test_norm_activation_oracle_properties():
    """NormActivation restatement: acts on the norm of every irrep copy only (so it commutes with any
    orthogonal map inside a copy), clamped copies stay finite and pass no gradient through the norm,
    fp64 gradcheck away from zero."""
    from oracle import e3ref as E
    from oracle import eigenn_ref as R

    irreps = E.Irreps("3x0e+2x1o+2x2e")
    m = E.NormActivation(irreps, R.ACTIVATION["e"]["ssp"], normalize=True, epsilon=1e-8)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, irreps.dim, dtype=torch.float64, generator=g)
    y = m(x)
    # an orthogonal map applied inside each copy commutes with the activation
    q1, _ = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64, generator=g))
    q2, _ = torch.linalg.qr(torch.randn(5, 5, dtype=torch.float64, generator=g))

    def rot(t):
        out = t.clone()
        out[:, 3:9] = (t[:, 3:9].reshape(-1, 2, 3) @ q1.T).reshape(-1, 6)
        out[:, 9:19] = (t[:, 9:19].reshape(-1, 2, 5) @ q2.T).reshape(-1, 10)
        return out

    assert torch.allclose(m(rot(x)), rot(y), atol=1e-12)
    # norms of the outputs are |ssp(n)|, independent of direction
    n_in = x[:, 3:6].norm(dim=1)
    assert torch.allclose(y[:, 3:6].norm(dim=1), R.ACTIVATION["e"]["ssp"](n_in).abs(), atol=1e-12)
    z = torch.zeros(2, irreps.dim, dtype=torch.float64, requires_grad=True)
    yz = m(z)
    (gz,) = torch.autograd.grad(yz.sum(), z)
    assert torch.isfinite(yz).all() and torch.isfinite(gz).all()
    assert torch.autograd.gradcheck(m, (x.clone().requires_grad_(True),))
