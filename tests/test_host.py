"""CPU: host logic of matten_amd (no kernel is launched): irreps algebra, that the product
modules derive exactly the irreps / instruction tables / parameter layout the oracle (hence the
reference) does, that libtfc.so loads and exports every symbol include/tfc.h declares, and the
library's host-side Clebsch-Gordan tables."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import matten_amd.nn as H
from matten_amd import _lib, so3
from matten_amd import keys as K
from matten_amd.irreps import Irrep, Irreps, tp_path_exists
from oracle import e3ref as e3
from oracle import eigenn_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- irreps
def test_irreps_algebra():
    a = Irreps("32x0o + 32x0e+16x1o")
    assert a.dim == 32 + 32 + 48 and a.num_irreps == 80 and str(a) == "32x0o+32x0e+16x1o"
    assert Irreps("0e") == Irreps("1x0e") and Irreps("1o") == "1x1o"
    assert Irreps("32x0e+24x0e") != Irreps("56x0e")
    assert str(Irreps("32x0e+24x0e+16x1o").simplify()) == "56x0e+16x1o"
    assert str(Irreps("4x0e+0x1o+2x0e").simplify()) == "4x0e+2x0e".replace("4x0e+2x0e", "6x0e")
    assert [str(i) for i in Irrep.of("1o") * Irrep.of("2e")] == ["1o", "2o", "3o"]
    s, p, inv = Irreps("4x1e+2x0e+3x1o+1x0o").sort()
    assert str(s) == "1x0o+2x0e+3x1o+4x1e" and p == (3, 1, 2, 0) and inv == (3, 1, 2, 0)
    assert "1o" in a and "2e" not in a and Irrep(0, 1) in a
    assert str(Irreps.spherical_harmonics(3)) == "1x0e+1x1o+1x2e+1x3o"
    assert a.offsets() == [0, 32, 64] and a.ls[:2] == [0, 0]
    assert tp_path_exists("32x0e", "0e+1o+2e", "1o") and not tp_path_exists("32x0e", "0e+1o+2e", "1e")
    for bad in ("3x", "x0e", "2y", "-1x0e", "1x-1e"):
        with pytest.raises((ValueError, TypeError)):
            Irreps(bad)

    class FakeE3nnIrreps:  # any object whose str() is an irreps string is accepted
        def __str__(self):
            return "8x0e+4x1o"

    assert Irreps(FakeE3nnIrreps()).dim == 20


def test_irreps_match_oracle_on_random_strings():
    g = np.random.default_rng(0)
    for _ in range(200):
        items = [(int(g.integers(0, 5)), (int(g.integers(0, 4)), int(g.choice([-1, 1])))) for _ in range(g.integers(1, 7))]
        a, b = Irreps(items), e3.Irreps(items)
        assert str(a) == str(b) and a.dim == b.dim and str(a.simplify()) == str(b.simplify())
        sa, sb = a.sort(), b.sort()
        assert str(sa[0]) == str(sb.irreps) and tuple(sa[1]) == tuple(sb.p) and tuple(sa[2]) == tuple(sb.inv)


# ----------------------------------------------------------------------------- CG
def test_cg_three_implementations_agree():
    for l1 in range(4):
        for l2 in range(4):
            for l3 in range(abs(l1 - l2), min(3, l1 + l2) + 1):
                ref = e3.wigner_3j(l1, l2, l3).numpy()
                assert np.abs(so3.real_cg(l1, l2, l3) - ref).max() < 1e-14
                lib = np.array(_lib.real_cg(l1, l2, l3)).reshape(ref.shape)  # C++ tables inside libtfc
                assert np.abs(lib - ref).max() < 1e-14
    L = _lib.lib()
    buf = (ctypes.c_double * 27)()
    assert L.tfc_real_cg(1, 1, 3, buf) == -1 and b"triangle" in L.tfc_last_error()
    assert L.tfc_real_cg(1, 1, 1, None) == -1


# ----------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "tfc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tfc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"libtfc.so does not export {missing}"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with include/tfc.h"
    assert _lib.lib().tfc_version() >= 100


def test_no_oracle_import_in_product():
    """The product path must not route through the oracle or any CPU fallback."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "matten_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


# ----------------------------------------------------------------------------- structure parity
CONFIGS = [
    (32, "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", 3),
    (64, "0e+1o+2e", "64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", 3),
    (128, "0e+1o+2e+3o", "128x0o+128x0e+128x1o+128x1e+128x2o+128x2e+128x3o+128x3e", 3),
    (8, "0e+1o", "4x0e+4x1o+4x1e+2x2e", 4),
]


@pytest.mark.parametrize("emb,sh,conv,n_layers", CONFIGS)
def test_product_layers_mirror_oracle_structure(emb, sh, conv, n_layers):
    irr = {K.NODE_FEATURES: f"{emb}x0e", K.NODE_ATTRS: f"{emb}x0e", K.EDGE_EMBEDDING: "8x0e", K.EDGE_ATTRS: sh, K.EDGE_INDEX: None}
    io, ih = dict(irr), dict(irr)
    kw = dict(message_kwargs=dict(fc_num_hidden_layers=2, fc_hidden_size=64), update_kwargs=dict(use_self_connection=True, avg_num_neighbors=9.0))
    for _ in range(n_layers):
        lo = R.PointConvMessagePassing(io, conv, **kw)
        lh = H.PointConvMessagePassing(ih, conv, **kw)
        assert {k: str(v) for k, v in lo.irreps_out.items()} == {k: str(v) for k, v in lh.irreps_out.items()}
        assert {k: str(v) for k, v in lo.irreps_in.items()} == {k: str(v) for k, v in lh.irreps_in.items()}
        assert lo.use_resnet == lh.use_resnet
        to, th = lo.message_fn.tp, lh.message_fn.tp
        assert str(to.irreps_mid) == str(th.irreps_mid)
        assert [(i.i_in1, i.i_in2, i.i_out) for i in to.tp.instructions] == th.tp.instructions
        assert to.tp.weight_numel == th.tp.weight_numel
        assert np.allclose(to.tp.coeffs, [p["coeff"] for p in th.tp.plan.paths])
        so, sh_ = lo.state_dict(), lh.state_dict()
        assert {k: tuple(v.shape) for k, v in so.items()} == {k: tuple(v.shape) for k, v in sh_.items()}
        lh.load_state_dict(so, strict=True)
        g = lh.equivariant_nonlinear.activation
        assert str(g.irreps_in) == str(lo.equivariant_nonlinear.irreps_in) and str(g.irreps_out) == str(lo.equivariant_nonlinear.irreps_out)
        io, ih = lo.irreps_out, lh.irreps_out


def test_reference_state_dict_keys():
    """Keys a reference checkpoint of this layer holds."""
    irr = {K.NODE_FEATURES: "8x0e", K.NODE_ATTRS: "8x0e", K.EDGE_EMBEDDING: "8x0e", K.EDGE_ATTRS: "0e+1o", K.EDGE_INDEX: None}
    layer = H.PointConvMessagePassing(irr, "8x0e+8x1o", message_kwargs=dict(fc_num_hidden_layers=2, fc_hidden_size=64))
    assert set(layer.state_dict()) == {
        "message_fn.linear_1.weight",
        "message_fn.tp.weight_nn.layer0.weight",
        "message_fn.tp.weight_nn.layer1.weight",
        "message_fn.tp.weight_nn.layer2.weight",
        "update_fn.linear_2.weight",
        "update_fn.self_connection.weight",
    }
    rb = H.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0})
    assert set(rb.state_dict()) == {"basis.bessel_weights", "cutoff.p", "cutoff.r_max"}
    assert torch.allclose(rb.basis.bessel_weights, torch.linspace(1, 8, 8) * torch.pi)


def test_constructor_error_conventions():
    with pytest.raises(ValueError, match="requires"):
        H.PointConvMessage({K.NODE_FEATURES: "4x0e"}, {K.NODE_FEATURES: "4x0e"})
    with pytest.raises(ValueError, match="type"):
        H.PointConvMessage({K.NODE_FEATURES: "4x0e", K.EDGE_EMBEDDING: "4x1o", K.EDGE_ATTRS: "0e", K.EDGE_INDEX: None}, {K.NODE_FEATURES: "4x0e"})
    with pytest.raises(AssertionError, match="produces no instructions"):
        H.UVUTensorProduct("4x0e", "1o", "4x0e", mlp_input_size=4)
    with pytest.raises(ValueError, match="Positions"):
        H.PointConvMessage({K.POSITIONS: "1e", K.NODE_FEATURES: "4x0e", K.EDGE_EMBEDDING: "4x0e", K.EDGE_ATTRS: "0e", K.EDGE_INDEX: None}, {K.NODE_FEATURES: "4x0e"})
    with pytest.raises(ValueError, match="unable to produce gates"):
        H.ActivationLayer("4x1o", "2e", "4x1o+4x2o")  # 1o x 2e reaches 1o,2o,3o but no scalar for the gates
    with pytest.raises(ValueError, match="activation_type"):
        H.ActivationLayer("4x0e", "0e+1o", "4x0e+4x1o", activation_type="nope")


def test_activation_layer_norm_branch_matches_oracle():
    """activation_type="norm" (eigenn/nn/utils.py:142-150): same irreps decisions as the oracle, and the
    layer built with it publishes the same irreps and parameter shapes."""
    for in1, in2, out in [("8x0e+4x1o", "0e+1o+2e", "8x0o+8x0e+4x1o+4x1e+2x2e"), ("4x0e", "0e+1o", "4x0e+4x1o+4x2e")]:
        a = H.ActivationLayer(in1, in2, out, activation_type="norm")
        b = R.ActivationLayer(in1, in2, out, activation_type="norm")
        assert str(a.irreps_in) == str(b.irreps_in) and str(a.irreps_out) == str(b.irreps_out) == str(a.irreps_in)
    irr = {K.NODE_FEATURES: "8x0e", K.NODE_ATTRS: "8x0e", K.EDGE_ATTRS: "0e+1o+2e", K.EDGE_EMBEDDING: "8x0e", K.EDGE_INDEX: None}
    kw = dict(conv_layer_irreps="8x0o+8x0e+4x1o+4x1e", message_kwargs={"fc_num_hidden_layers": 1, "fc_hidden_size": 8},
              update_kwargs={"use_self_connection": True, "avg_num_neighbors": 10.0}, activation_type="norm")
    la, lb = H.PointConvMessagePassing(irreps_in=dict(irr), **kw), R.PointConvMessagePassing(irreps_in=dict(irr), **kw)
    assert {k: str(v) for k, v in la.irreps_out.items()} == {k: str(v) for k, v in lb.irreps_out.items()}
    assert {k: tuple(v.shape) for k, v in la.state_dict().items()} == {k: tuple(v.shape) for k, v in lb.state_dict().items()}


def test_reference_constructor_smoke_test():
    """tests/nn/test_convolution.py:8-21 of the reference: the legacy PointConv constructs."""
    conv = H.PointConv(
        irreps_in={K.NODE_FEATURES: Irreps("4x1e+2x0e"), K.NODE_ATTRS: Irreps("4x1e+2x0e"), K.EDGE_ATTRS: Irreps("4x1e+2x0e"),
                   K.EDGE_EMBEDDING: Irreps("4x0e")},
        irreps_out={K.NODE_FEATURES: Irreps("8x1e+4x0e")},
    )
    ref = R.PointConv(
        irreps_in={K.NODE_FEATURES: "4x1e+2x0e", K.NODE_ATTRS: "4x1e+2x0e", K.EDGE_ATTRS: "4x1e+2x0e", K.EDGE_EMBEDDING: "4x0e"},
        irreps_out={K.NODE_FEATURES: "8x1e+4x0e"},
    )
    assert {k: tuple(v.shape) for k, v in conv.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}


def test_ops_refuse_cpu_tensors():
    from matten_amd.nn.e3 import Linear

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Linear("4x0e", "4x0e")(torch.randn(3, 4))


@pytest.mark.parametrize("with_hint", [True, False])
def test_attr_factors_host_logic(with_hint):
    """The factorisation a = C R behind the low-rank self connection (matten_amd/ops.py
    attr_factors) is host-side torch: exact reconstruction, refusal when it would not pay, and the
    cache keyed on (address, version) noticing an in-place update."""
    from matten_amd import ops

    g = torch.Generator().manual_seed(0)
    table = torch.randn(5, 32, generator=g)
    idx = torch.randint(0, 5, (100,), generator=g)
    a = table[idx].contiguous()
    f = ops.attr_factors(a, idx if with_hint else None)
    assert f is not None
    Cm, Rm = f.C, f.R
    assert f.promised == with_hint and torch.equal(f.inv_count[:5] * Cm.sum(0)[:5], torch.ones(5))
    assert Cm.shape == (100, 8) and Rm.shape == (8, 32)  # 5 classes padded to a multiple of 4
    assert torch.equal(Cm @ Rm, a) and torch.equal(Cm.sum(1), torch.ones(100))
    assert ops.attr_factors(a, idx if with_hint else None) is f  # cached
    a.mul_(2.0)  # version bump: factors are rebuilt
    f2 = ops.attr_factors(a, idx if with_hint else None)
    assert f2 is not f and torch.equal(f2.C @ f2.R, a)
    # too few rows, too narrow, or too many distinct rows: dense route
    assert ops.attr_factors(a[:10].contiguous(), idx[:10] if with_hint else None) is None
    assert ops.attr_factors(torch.randn(100, 4), None) is None
    many = torch.randn(100, 32, generator=g)
    assert ops.attr_factors(many, torch.arange(100) if with_hint else None) is None
    if not with_hint:
        nan_rows = a.clone()
        nan_rows[3, 0] = float("nan")
        assert ops.attr_factors(nan_rows) is None  # NaN rows never verify as copies


def test_every_cache_of_the_ops_is_registered():
    """ops.clear_caches() walks one registry; a module-level dict that is not in it must be on the short list of
    tables that hold no device tensors (a forgotten cache brings the graph-capture failure back)."""
    from matten_amd import ops

    registered = {id(d) for d in ops._TENSOR_CACHES}
    no_tensors = {"_FCTP_WS", "_SIDE_STREAMS"}  # byte counts; stream objects
    for name, val in vars(ops).items():
        if isinstance(val, dict) and name.startswith("_") and name.upper() == name and not name.startswith("__"):
            assert id(val) in registered or name in no_tensors, name
    assert len(registered) >= 7


def test_packed_batch_is_one_buffer_and_refuses_another_layout():
    import torch

    from matten_amd.graphs import PackedBatch

    g = torch.Generator().manual_seed(3)
    def batch(e):
        return {"pos": torch.randn(5, 3, generator=g), "edge_index": torch.randint(0, 5, (2, e), generator=g),
                "cell": torch.randn(1, 3, 3, generator=g), "flag": torch.tensor([1, 0, 1], dtype=torch.uint8)}

    a, b = batch(7), batch(7)
    static = PackedBatch(a)
    for k, v in a.items():
        assert torch.equal(static.tensors[k], v) and static.tensors[k].dtype == v.dtype
        assert static.tensors[k].data_ptr() % 16 == 0
    before = {k: (v.data_ptr(), v._version) for k, v in static.tensors.items()}
    static.load(PackedBatch(b, like=static))
    for k, v in b.items():
        assert torch.equal(static.tensors[k], v)
        assert static.tensors[k].data_ptr() == before[k][0]       # same addresses: a captured step keeps reading them
        assert static.tensors[k]._version != before[k][1]         # and the version-keyed caches of the eager path miss
    import pytest

    with pytest.raises(ValueError, match="differ"):
        PackedBatch(batch(8), like=static)
