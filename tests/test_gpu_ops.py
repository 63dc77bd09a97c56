"""-m gpu: every libtfc kernel, called through the C ABI (ctypes), against the CPU oracle on
the same seeded inputs.  Tolerance: 1e-5 relative (fp32) to the tensor's max-norm, as
BASELINE.json's north_star states; integer orderings must be bit-exact."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    s = ref.abs().max().item()
    return (a - ref).abs().max().item() / (s if s > 0 else 1.0)


def random_graph(n_nodes, n_edges, seed, device):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n_nodes, (2, n_edges), generator=g)
    return ei.to(device)


# --------------------------------------------------------------------------------------- csr
@pytest.mark.parametrize("n_nodes,n_edges", [(1, 1), (7, 0), (50, 400), (3000, 90000), (5, 3000)])
def test_csr_build_bit_exact(device, n_nodes, n_edges):
    from matten_amd import ops

    ei = random_graph(n_nodes, n_edges, 1, device)
    o = ops.EdgeOrdering(ei, n_nodes)
    o.validate()
    for key_row, (rp, eid, nbr) in ((1, o.by_dst), (0, o.by_src)):
        key = ei[key_row].cpu()
        other = ei[1 - key_row].cpu()
        order = torch.sort(key, stable=True).indices
        counts = torch.bincount(key, minlength=n_nodes)
        rp_ref = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
        assert torch.equal(rp.cpu().long(), rp_ref)
        if n_edges:
            assert torch.equal(eid.cpu().long()[:n_edges], order)
            assert torch.equal(nbr.cpu().long()[:n_edges], other[order])


def test_csr_flags_bad_index(device):
    from matten_amd import ops

    ops.check_edge_orderings(wait=True)  # nothing pending from earlier tests
    ei = torch.tensor([[0, 1, 5], [1, 0, 1]], device=device)
    o = ops.EdgeOrdering(ei, 3)
    with pytest.raises(IndexError):
        o.validate()
    # the deferred report: the flag reaches the host behind the build, and the next ordering that is built (or
    # an explicit check) raises - the reference's gather would have raised at once (eigenn/nn/point_conv.py:82)
    torch.cuda.synchronize()
    with pytest.raises(IndexError, match="outside"):
        ops.edge_ordering(torch.tensor([[0, 1], [1, 0]], device=device), 2)
    ops.check_edge_orderings(wait=True)  # reported once, then forgotten
    good = ops.edge_ordering(torch.tensor([[0, 1], [1, 0]], device=device), 2)
    good.validate()


# --------------------------------------------------------------------------------------- gemm
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (100, 70, 37), (64, 64, 16), (300, 130, 8), (1000, 1728, 64), (64, 200, 5000), (4096, 64, 333)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_plain(device, m, n, k, layout):
    from matten_amd import ops

    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    A = torch.randn(m, k, generator=g)
    B = torch.randn(k, n, generator=g)
    ref = (A.double() @ B.double()) * 0.37
    if layout == "nn":
        a, b = A.contiguous().to(device), B.contiguous().to(device)
        a_r, a_k, b_k, b_j = ops._lin(k), ops._lin(1), ops._lin(n), ops._lin(1)
    elif layout == "nt":
        a, b = A.contiguous().to(device), B.t().contiguous().to(device)
        a_r, a_k, b_k, b_j = ops._lin(k), ops._lin(1), ops._lin(1), ops._lin(k)
    else:
        a, b = A.t().contiguous().to(device), B.contiguous().to(device)
        a_r, a_k, b_k, b_j = ops._lin(1), ops._lin(m), ops._lin(n), ops._lin(1)
    c = torch.full((m, n), float("nan"), device=device)
    ops._gemm(m, n, k, ops._p(a), ops._p(b), ops._p(c), a_r, a_k, b_k, b_j, ops._lin(n), ops._lin(1), alpha=0.37)
    assert rel(c, ref) < TOL
    # accumulate on top
    ops._gemm(m, n, k, ops._p(a), ops._p(b), ops._p(c), a_r, a_k, b_k, b_j, ops._lin(n), ops._lin(1), alpha=0.37, accumulate=True)
    assert rel(c, 2 * ref) < TOL


# --------------------------------------------------------------------------------------- e3 blocks
def _pair_state(oracle_mod, hip_mod):
    hip_mod.load_state_dict({k: v.float() for k, v in oracle_mod.state_dict().items()}, strict=True)


@pytest.mark.parametrize(
    "irreps_in,irreps_out",
    [
        ("32x0e", "32x0e"),
        ("32x0e+16x1o+8x2e", "32x0e+16x1o+8x2e"),
        ("56x0e+72x1o+24x1e+24x2o+64x2e", "80x0e+16x1o+16x1e+8x2o+8x2e"),
        ("4x1e+2x0e", "8x1e+4x0e"),
        ("8x0e+8x0e+3x1o", "5x0e+2x1o+2x1o+3x2e"),
        ("130x0e+70x3o", "65x0e+100x3o"),
    ],
)
def test_linear_vs_oracle(device, irreps_in, irreps_out):
    from matten_amd.nn.e3 import Linear
    from oracle import e3ref

    torch.manual_seed(0)
    o = e3ref.Linear(irreps_in, irreps_out).double()
    h = Linear(irreps_in, irreps_out)
    _pair_state(o, h)
    h = h.to(device)
    n = 77
    x = torch.randn(n, o.irreps_in.dim, dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    y, yg = o(x), h(xg)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, [x, o.weight], go)
    gxg, gwg = torch.autograd.grad(yg, [xg, h.weight], go.float().to(device))
    assert rel(gxg, gx) < TOL
    assert rel(gwg, gw) < TOL


@pytest.mark.parametrize(
    "in1,in2,out",
    [
        ("32x0e", "32x0e", "56x0e+16x1o+8x2e"),
        ("32x0e+16x1o+8x2e", "32x0e", "80x0e+16x1o+16x1e+8x2o+8x2e"),
        ("64x0e+64x1o+64x1e+64x2o+64x2e", "64x0e", "64x0o+256x0e+64x1o+64x1e+64x2o+64x2e"),
        ("5x0e+3x1o+2x3o", "7x0e", "4x0e+6x1o+3x3o"),
    ],
)
def test_fctp_vs_oracle(device, in1, in2, out):
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    torch.manual_seed(1)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    _pair_state(o, h)
    h = h.to(device)
    n = 41
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    a = torch.randn(n, o.irreps_in2.dim, dtype=torch.float64, requires_grad=True)
    base = torch.randn(n, o.irreps_out.dim, dtype=torch.float64, requires_grad=True)
    xg, ag, bg = (t.detach().float().to(device).requires_grad_(True) for t in (x, a, base))
    y = base + o(x, a)
    yg = h(xg, ag, base=bg * 1.0)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x, a, o.weight, base], go)
    got = torch.autograd.grad(yg, [xg, ag, h.weight, bg], go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL
    # without base
    y2, y2g = o(x, a), h(xg, ag)
    assert rel(y2g, y2) < TOL


@pytest.mark.parametrize("n_distinct", [1, 3, 5, 17])
def test_fctp_low_rank_route_vs_oracle(device, n_distinct):
    """Node attributes that are copies of a few distinct rows (a species embedding) take the
    shorter contraction (ops.attr_factors); every gradient, the per-node one of the attributes
    included, must still be the dense one."""
    from matten_amd import ops
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    in1, in2, out = "16x0e+8x1o+8x2e", "32x0e", "16x0o+40x0e+8x1o+8x1e+8x2o+8x2e"
    torch.manual_seed(11)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    _pair_state(o, h)
    h = h.to(device)
    n = 203
    table = torch.randn(n_distinct, 32, dtype=torch.float64)
    idx = torch.randint(0, n_distinct, (n,))
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    a = table[idx].clone().requires_grad_(True)
    base = torch.randn(n, o.irreps_out.dim, dtype=torch.float64, requires_grad=True)
    xg, ag, bg = (t.detach().float().to(device).requires_grad_(True) for t in (x, a, base))
    f = ops.attr_factors(ag)
    if 2 * max(4, (n_distinct + 3) // 4 * 4) <= 32:
        assert f is not None and not f.promised and torch.equal(f.C @ f.R, ag.detach())
    else:
        assert f is None
    y = base + o(x, a)
    yg = h(xg, ag, base=bg * 1.0)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x, a, o.weight, base], go)
    got = torch.autograd.grad(yg, [xg, ag, h.weight, bg], go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL


@pytest.mark.parametrize("n_distinct", [1, 4, 7])
def test_fctp_promised_classes_project_attr_gradient(device, n_distinct):
    """With keys.NODE_ATTRS_CLASSES supplied the caller promises a = table[classes]; the attribute
    gradient is then returned projected onto such tensors: every row gets the mean of the dense
    gradient over its class, so the table behind the lookup receives exactly the dense gradient.
    All other results are unchanged."""
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    in1, in2, out = "16x0e+8x1o+8x2e", "32x0e", "16x0o+40x0e+8x1o+8x1e+8x2o+8x2e"
    torch.manual_seed(12)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    _pair_state(o, h)
    h = h.to(device)
    n = 150
    table = torch.randn(n_distinct, 32, dtype=torch.float64, requires_grad=True)
    idx = torch.randint(0, n_distinct, (n,))
    idx[:n_distinct] = torch.arange(n_distinct)
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    a = table[idx]
    a.retain_grad()
    y = o(x, a)
    go = torch.randn_like(y)
    gx, gt_, gw = torch.autograd.grad(y, [x, table, o.weight], go, retain_graph=True)
    (ga,) = torch.autograd.grad(y, [a], go)
    tg = table.detach().float().to(device).requires_grad_(True)
    xg = x.detach().float().to(device).requires_grad_(True)
    idx_d = idx.to(device)
    ag = tg[idx_d]
    yg = h(xg, ag, classes=idx_d)
    assert rel(yg, y) < TOL
    gxg, gtg, gwg, gag = torch.autograd.grad(yg, [xg, tg, h.weight, ag], go.float().to(device))
    assert rel(gxg, gx) < TOL and rel(gwg, gw) < TOL
    assert rel(gtg, gt_) < TOL  # the table gradient is the dense one
    onehot = torch.nn.functional.one_hot(idx, n_distinct).double()
    class_mean = (onehot / onehot.sum(0)) @ (onehot.T @ ga)
    assert rel(gag, class_mean) < TOL


@pytest.mark.parametrize("hs", [[8, 64, 64, 96], [8, 64, 384], [4, 8, 24], [8, 64, 64, 1728]])
def test_radial_mlp_vs_oracle(device, hs):
    from matten_amd.nn.e3 import FullyConnectedNet
    from oracle import e3ref
    from oracle.eigenn_ref import ssp

    torch.manual_seed(2)
    o = e3ref.FullyConnectedNet(hs, act=ssp, act_key="ssp").double()
    h = FullyConnectedNet(hs, act="ssp")
    _pair_state(o, h)
    h = h.to(device)
    e = 1234
    x = torch.randn(e, hs[0], dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    y, yg = o(x), h(xg)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x] + list(o.parameters()), go)
    got = torch.autograd.grad(yg, [xg] + list(h.parameters()), go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL


@pytest.mark.parametrize("act", ["silu", "sigmoid"])
def test_radial_mlp_other_activations_vs_oracle(device, act):
    """mlp_activation other than the shifted softplus (eigenn/nn/utils.py:182,252), forward, backward and through
    UVUTensorProduct's constructor argument (by name and as the callable the reference passes)."""
    from matten_amd.nn.e3 import FullyConnectedNet
    from matten_amd.nn.utils import UVUTensorProduct
    from oracle import e3ref

    fn = {"silu": torch.nn.functional.silu, "sigmoid": torch.sigmoid}[act]
    hs = [8, 64, 64, 96]
    torch.manual_seed(4)
    o = e3ref.FullyConnectedNet(hs, act=fn, act_key=act).double()
    h = FullyConnectedNet(hs, act=act)
    _pair_state(o, h)
    h = h.to(device)
    x = torch.randn(777, hs[0], dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    y, yg = o(x), h(xg)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x] + list(o.parameters()), go)
    got = torch.autograd.grad(yg, [xg] + list(h.parameters()), go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL
    for arg in (act, fn):
        tp = UVUTensorProduct("8x0e+4x1o", "0e+1o+2e", "8x0e+4x1o", mlp_input_size=4, mlp_activation=arg)
        assert tp.weight_nn.act == act
    with pytest.raises(ValueError):
        UVUTensorProduct("8x0e+4x1o", "0e+1o+2e", "8x0e+4x1o", mlp_input_size=4, mlp_activation="relu6")


@pytest.mark.parametrize(
    "in1,in2,out",
    [
        ("4x1e+2x0e", "4x1e+2x0e", "8x1e+4x0e"),
        ("3x0e+2x1o+2x2e", "2x1o+1x2e+2x0e", "4x0e+3x1o+2x1e+2x2e+2x3o"),
    ],
)
def test_fctp_general_second_input_vs_oracle(device, in1, in2, out):
    """FullyConnectedTensorProduct with a NON-scalar second input (the reference's constructor test gives the node
    attributes 4x1e+2x0e): forward, input and weight gradients, and the accumulation onto a base."""
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    torch.manual_seed(6)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    assert h._general and h.weight_numel == o.weight.numel()
    h.load_state_dict({"weight": o.weight.detach().float()}, strict=False)
    h = h.to(device)
    n = 57
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    a = torch.randn(n, o.irreps_in2.dim, dtype=torch.float64, requires_grad=True)
    base = torch.randn(n, o.irreps_out.dim, dtype=torch.float64)
    xg, ag = (t.detach().float().to(device).requires_grad_(True) for t in (x, a))
    y = o(x, a) + base
    yg = h(xg, ag, base=base.float().to(device))
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x, a, o.weight], go)
    got = torch.autograd.grad(yg, [xg, ag, h.weight], go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL


def test_uvu_second_input_multiplicity_vs_oracle(device):
    """'uvu' tensor product whose second input has blocks of multiplicity > 1 (weight blocks [mul_1, mul_2]),
    fused with gather and per-receiver sum, against the oracle's unfused op order."""
    from matten_amd.nn.utils import UVUTensorProduct
    from oracle import eigenn_ref as R

    torch.manual_seed(8)
    args = ("4x1e+2x0e+3x1o", "4x1e+2x0e+2x2e", "8x1e+4x0e+3x1o+2x2o")
    o = R.UVUTensorProduct(*args, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1).double()
    h = UVUTensorProduct(*args, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1)
    assert h.tp._expanded and str(h.irreps_out) == str(o.irreps_out)
    _pair_state(o, h)
    h = h.to(device)
    n, e = 19, 140
    ei = random_graph(n, e, 61, device)
    x = torch.randn(n, h.tp.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    y = torch.randn(e, h.tp.irreps_in2.dim, dtype=torch.float64, requires_grad=True)
    emb = torch.randn(e, 4, dtype=torch.float64, requires_grad=True)
    xg, yg, eg = (t.detach().float().to(device).requires_grad_(True) for t in (x, y, emb))
    from matten_amd import ops

    msg = o(x[ei[0].cpu()], y, emb)
    ref_out = torch.zeros(n, msg.shape[1], dtype=torch.float64).index_add_(0, ei[1].cpu(), msg)
    got_out = h(xg, yg, eg, ordering=ops.edge_ordering(ei, n))
    assert rel(got_out, ref_out) < TOL
    go = torch.randn_like(ref_out)
    ref = torch.autograd.grad(ref_out, [x, y, emb] + list(o.parameters()), go)
    got = torch.autograd.grad(got_out, [xg, yg, eg] + list(h.parameters()), go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL


@pytest.mark.parametrize(
    "scalars,acts_s,gates,acts_g,gated",
    [
        ("32x0e", ["ssp"], "24x0e", ["ssp"], "16x1o+8x2e"),
        ("32x0o+32x0e", ["tanh", "ssp"], "48x0e", ["ssp"], "16x1o+16x1e+8x2o+8x2e"),
        ("8x0o", ["tanh"], "6x0o", ["abs"], "4x1o+2x2e"),
        ("5x0e", ["silu"], "", [], ""),
        ("3x0e", ["sigmoid"], "4x0o", ["tanh"], "4x1e"),
    ],
)
def test_gate_vs_oracle(device, scalars, acts_s, gates, acts_g, gated):
    from matten_amd.nn.e3 import Gate
    from oracle import e3ref
    from oracle.eigenn_ref import ACTIVATION

    fn = {**ACTIVATION["e"], **ACTIVATION["o"]}
    o = e3ref.Gate(scalars, [fn[a] for a in acts_s], gates, [fn[a] for a in acts_g], gated, keys_scalars=acts_s, keys_gates=acts_g)
    h = Gate(scalars, acts_s, gates, acts_g, gated)
    assert str(h.irreps_in) == str(o.irreps_in) and str(h.irreps_out) == str(o.irreps_out)
    torch.manual_seed(3)
    x = torch.randn(93, o.irreps_in.dim, dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    y, yg = o(x), h(xg)
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, [x], go)
    (gxg,) = torch.autograd.grad(yg, [xg], go.float().to(device))
    assert rel(gxg, gx) < TOL


# --------------------------------------------------------------------------------------- conv
@pytest.mark.parametrize(
    "in1,sh,out,n_nodes,n_edges",
    [
        ("8x0e", "0e+1o+2e", "8x0o+8x0e+4x1o+4x1e+2x2o+2x2e", 10, 120),
        ("32x0e+16x1o+8x2e", "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", 30, 500),
        ("32x0e+16x1o+16x1e+8x2o+8x2e", "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", 30, 700),
        ("64x0e+64x1o+64x1e+64x2o+64x2e", "0e+1o+2e", "64x0o+64x0e+64x1o+64x1e+64x2o+64x2e", 20, 300),
        ("96x0e+70x1o+130x2e", "0e+1o+2e", "96x0e+70x1o+130x2e+70x1e", 9, 100),
        ("16x0e+16x1o+8x2e+4x3o", "0e+1o+2e+3o", "16x0o+16x0e+16x1o+16x1e+8x2o+8x2e+4x3o+4x3e", 12, 200),
        ("8x0e+8x3o+8x3e", "0e+1o+2e+3o", "8x0e+8x1o+8x2e+8x3o+8x3e", 6, 1),
        # receivers with more than 64 / more than 128 incoming edges (the forward kernel's chunked edge loop, its
        # tail chunk and the remainder of its unrolled groups), channel counts that are not multiples of 64
        ("96x0e+70x1o+130x2e", "0e+1o+2e", "96x0e+70x1o+130x2e+70x1e", 5, 3000),
        ("40x0e+24x1o+24x1e+10x2e", "0e+1o+2e", "40x0o+40x0e+24x1o+24x1e+10x2o+10x2e", 3, 451),
        ("20x0e+12x1o+6x2e+3x3o", "0e+1o+2e+3o", "20x0o+20x0e+12x1o+12x1e+6x2o+6x2e+3x3o+3x3e", 4, 333),
        ("64x0e+64x1o", "0e+1o+2e", "64x0e+64x1o+64x1e+64x2e", 2, 130),
    ],
)
@pytest.mark.parametrize("need_dy", [True, False])
def test_conv_vs_oracle(device, in1, sh, out, n_nodes, n_edges, need_dy):
    """gather -> uvu tensor product -> scatter_add -> scale, forward and all three gradients."""
    from matten_amd import ops
    from matten_amd.nn.utils import UVUTensorProduct
    from oracle import eigenn_ref as R

    torch.manual_seed(4)
    o = R.UVUTensorProduct(in1, sh, out, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1)
    h = UVUTensorProduct(in1, sh, out, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1)
    assert str(h.irreps_out) == str(o.irreps_out)
    ei = random_graph(n_nodes, n_edges, 5, "cpu")
    x = torch.randn(n_nodes, o.tp.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    v = torch.randn(n_edges, 3, dtype=torch.float64)
    y = R.spherical_harmonics(sh, v, True, "component").requires_grad_(True)
    w = torch.randn(n_edges, o.tp.weight_numel, dtype=torch.float64, requires_grad=True)
    scale = 0.41
    msg = o.tp(x[ei[0]], y, w)
    ref = R.scatter_add(msg, ei[1], n_nodes) * scale
    go = torch.randn_like(ref)
    gref = torch.autograd.grad(ref, [x, y, w], go, retain_graph=True)

    xg, yg, wg = (t.detach().float().to(device).requires_grad_(True) for t in (x, y, w))
    if not need_dy:
        yg = yg.detach()
    ordering = ops.EdgeOrdering(ei.to(device), n_nodes)
    got = h.tp(xg, yg, wg, ordering, scale)
    assert rel(got, ref) < TOL
    ins = [xg, yg, wg] if need_dy else [xg, wg]
    ggot = torch.autograd.grad(got, ins, go.float().to(device))
    refs = gref if need_dy else (gref[0], gref[2])
    for r, g_ in zip(refs, ggot):
        assert rel(g_, r) < TOL
    # the unfused per-edge route gives the reference's EDGE_MESSAGE tensor
    per_edge = h.tp(xg, yg, wg, ordering, 1.0, per_edge=True)
    assert rel(per_edge, msg) < TOL
    gm = torch.randn_like(msg)
    gref2 = torch.autograd.grad(msg, [x, w], gm)
    ggot2 = torch.autograd.grad(per_edge, [xg, wg], gm.float().to(device))
    for r, g_ in zip(gref2, ggot2):
        assert rel(g_, r) < TOL


def test_conv_deterministic(device):
    from matten_amd import ops
    from matten_amd.nn.utils import UVUTensorProduct

    torch.manual_seed(6)
    h = UVUTensorProduct("32x0e+16x1o+8x2e", "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", mlp_input_size=4)
    n, e = 40, 2000
    ei = random_graph(n, e, 7, device)
    x = torch.randn(n, 120, device=device)
    y = torch.randn(e, 9, device=device)
    w = torch.randn(e, h.tp.weight_numel, device=device)
    outs = [h.tp(x, y, w, ops.EdgeOrdering(ei, n), 1.0) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_scatter_add_route(device):
    from matten_amd.nn.utils import scatter_add

    torch.manual_seed(8)
    src = torch.randn(500, 37, dtype=torch.float64, requires_grad=True)
    idx = torch.randint(0, 20, (500,))
    ref = torch.zeros(25, 37, dtype=torch.float64).index_add_(0, idx, src)
    sg = src.detach().float().to(device).requires_grad_(True)
    got = scatter_add(sg, idx.to(device), 25)
    assert rel(got, ref) < TOL
    go = torch.randn_like(ref)
    (gr,) = torch.autograd.grad(ref, [src], go)
    (gg,) = torch.autograd.grad(got, [sg], go.float().to(device))
    assert rel(gg, gr) < TOL


# --------------------------------------------------------------------------------------- edge encoders
@pytest.mark.parametrize("lmax", [0, 1, 2, 3])
@pytest.mark.parametrize("batched_cell", [True, False])
def test_edge_encoders_vs_oracle(device, lmax, batched_cell):
    import matten_amd.nn as H
    from matten_amd import keys as K
    from matten_amd.data.crystal import synthetic_batch
    from oracle import eigenn_ref as R

    batch, _ = synthetic_batch(2, 2.6, 2 if batched_cell else 1, seed0=77)
    torch.manual_seed(9)

    def run(M, dev, dtype):
        data = {k: (v.to(dtype) if v.is_floating_point() else v).to(dev) for k, v in batch.items()}
        data[K.POSITIONS] = data[K.POSITIONS].clone().requires_grad_(True)
        sh = M.SphericalHarmonicEdgeAttrs(irreps_edge_sh=lmax)
        rb = M.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0}).to(dev, dtype)
        data = rb(sh(data))
        g = torch.Generator().manual_seed(10)
        ga = torch.randn(data[K.EDGE_ATTRS].shape, generator=g, dtype=torch.float64).to(dev, dtype)
        ge = torch.randn(data[K.EDGE_EMBEDDING].shape, generator=g, dtype=torch.float64).to(dev, dtype)
        gv = torch.randn(data[K.EDGE_VECTORS].shape, generator=g, dtype=torch.float64).to(dev, dtype)
        gl = torch.randn(data[K.EDGE_LENGTH].shape, generator=g, dtype=torch.float64).to(dev, dtype)
        loss = (data[K.EDGE_ATTRS] * ga).sum() + (data[K.EDGE_EMBEDDING] * ge).sum() + (data[K.EDGE_VECTORS] * gv).sum() + (data[K.EDGE_LENGTH] * gl).sum()
        dpos, dbw = torch.autograd.grad(loss, [data[K.POSITIONS], rb.basis.bessel_weights])
        return data, dpos, dbw

    dr, pr, br = run(R, "cpu", torch.float64)
    dg, pg, bg = run(H, device, torch.float32)
    for key in (K.EDGE_VECTORS, K.EDGE_LENGTH, K.EDGE_ATTRS, K.EDGE_EMBEDDING):
        assert rel(dg[key], dr[key]) < TOL, key
    assert rel(pg, pr) < TOL
    assert rel(bg, br) < TOL


def test_c_abi_rejects_bad_arguments(device):
    """Error convention of the C ABI: negative code + message, no launch."""
    import ctypes as C

    from matten_amd import _lib

    L = _lib.lib()
    bad = (_lib.Path * 1)()
    bad[0].l1, bad[0].l2, bad[0].l3, bad[0].mul = 5, 0, 5, 4
    out = C.c_void_p()
    rc = L.tfc_conv_plan_create(bad, 1, 44, 1, 44, 4, C.byref(out))
    assert rc == -2 and b"outside" in L.tfc_last_error()
    bad[0].l1, bad[0].l3 = 1, 3  # triangle violation with l2 = 0
    rc = L.tfc_conv_plan_create(bad, 1, 44, 1, 44, 4, C.byref(out))
    assert rc == -1
    assert L.tfc_conv_fwd(None, None, None, None, None, None, None, 1, 1, 1.0, None, None, None) == -1


# --------------------------------------------------------------------------------------- newer entry points
def test_conv_one_call_entry_points_match_m_route(device):
    """tfc_conv_fwd / tfc_conv_bwd (M built inside, workspace argument) against the tfc_conv_m_build +
    tfc_conv_fwd_m / tfc_conv_bwd_m route the modules take: same kernels, so bit-identical."""
    import ctypes as C

    from matten_amd import _lib, ops
    from matten_amd.nn.utils import UVUTensorProduct

    torch.manual_seed(20)
    h = UVUTensorProduct("32x0e+16x1o+8x2e", "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", mlp_input_size=4)
    plan = h.tp.plan
    n, e = 37, 900
    ei = random_graph(n, e, 21, device)
    o = ops.EdgeOrdering(ei, n)
    x = torch.randn(n, plan.d_in, device=device, requires_grad=True)
    y = torch.randn(e, plan.d_sh, device=device, requires_grad=True)
    w = torch.randn(e, plan.w_numel, device=device, requires_grad=True)
    out = ops.conv(x, y, w, plan, o, 0.3)
    go = torch.randn_like(out)
    dx, dy, dw = torch.autograd.grad(out, [x, y, w], go)

    L = _lib.lib()
    hd = plan.handle(x.device)
    st = ops._stream()
    ws = plan.workspace(x.device, e, True)
    out2 = torch.full_like(out, float("nan"))
    rp, eid, nbr = o.by_dst
    P = ops._ptr
    _lib.check(L.tfc_conv_fwd(hd, P(x), P(y), P(w), P(rp), P(eid), P(nbr), n, e, 0.3, P(out2), P(ws), st))
    assert torch.equal(out2, out)
    rp, eid, nbr = o.by_src
    dx2, dy2, dw2 = (torch.full_like(t, float("nan")) for t in (dx, dy, dw))
    _lib.check(L.tfc_conv_bwd(hd, P(x), P(y), P(w), P(go), P(rp), P(eid), P(nbr), n, e, 0.3, P(dx2), P(dw2), P(dy2), P(ws), st))
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw) and torch.equal(dy2, dy)
    # plans that lay M out the same way share a signature; a different set of (l1, l2, l3) does not
    h2 = UVUTensorProduct("8x0e+4x1o+2x2e", "0e+1o+2e", "8x0o+8x0e+4x1o+4x1e+2x2o+2x2e", mlp_input_size=4)
    h3 = UVUTensorProduct("8x0e+4x1o", "0e+1o+2e", "8x0e+4x1o", mlp_input_size=4)
    s1, s2, s3 = (int(L.tfc_conv_m_signature(p.tp.plan.handle(x.device))) for p in (h, h2, h3))
    assert s1 == s2 and s1 != s3
    assert L.tfc_conv_m_floats(hd, e) == L.tfc_conv_m_floats(h2.tp.plan.handle(x.device), e) > 0


@pytest.mark.parametrize("n_rows", [1, 63, 200])
def test_ir_transpose(device, n_rows):
    """out[off*N + (m*N + n)*mul + u] = x[n, off + u*dim + m] for each (off, mul, dim) block; columns no
    block covers are left alone."""
    from matten_amd import ops

    blocks = [(0, 5, 1), (5, 3, 3), (20, 7, 5), (55, 2, 7), (69, 64, 3)]  # gap at [14, 20)
    d_row = 69 + 64 * 3
    g = torch.Generator().manual_seed(30 + n_rows)
    x = torch.randn(n_rows, d_row, generator=g).to(device)
    got = ops._ir_transpose(x, blocks).cpu()
    xc = x.cpu()
    for off, mul, dim in blocks:
        ref = xc[:, off : off + mul * dim].reshape(n_rows, mul, dim).permute(2, 0, 1).contiguous().reshape(-1)
        assert torch.equal(got[off * n_rows : off * n_rows + ref.numel()], ref), (off, mul, dim)


def test_ir_transpose_rejects_bad_blocks(device):
    from matten_amd import _lib, ops

    x = torch.randn(4, 10, device=device)
    out = torch.empty(40, device=device)
    arr = (_lib.IrBlock * 1)()
    arr[0].off, arr[0].mul, arr[0].dim = 6, 2, 3  # runs past the row
    assert _lib.lib().tfc_ir_transpose(ops._ptr(x), 4, 10, arr, 1, ops._ptr(out), ops._stream()) < 0


def test_gemm_group_mixed_shapes(device):
    """One tfc_gemm_group call over 19 problems that walk their operands in different ways (plain,
    transposed, two-factor A, split K): each must equal its own product."""
    from matten_amd import ops

    g = torch.Generator().manual_seed(40)
    descs, checks, keep = [], [], []
    for i in range(19):
        m, n, k = [(70, 33, 20), (128, 64, 512), (9, 200, 3000), (300, 12, 64)][i % 4]
        m += i
        A = torch.randn(m, k, generator=g)
        B = torch.randn(k, n, generator=g)
        kind = i % 3
        c = torch.full((m, n), float("nan"), device=device)
        if kind == 0:
            a, b = A.to(device), B.to(device)
            descs.append(ops._desc(m, n, k, ops._p(a), ops._p(b), ops._p(c), ops._lin(k), ops._lin(1), ops._lin(n), ops._lin(1), ops._lin(n), ops._lin(1), alpha=0.5))
            ref = 0.5 * (A.double() @ B.double())
        elif kind == 1:
            a, b = A.t().contiguous().to(device), B.t().contiguous().to(device)
            descs.append(ops._desc(m, n, k, ops._p(a), ops._p(b), ops._p(c), ops._lin(1), ops._lin(m), ops._lin(1), ops._lin(k), ops._lin(n), ops._lin(1)))
            ref = A.double() @ B.double()
        else:
            # A[r, t] = a1[r, t] * a2[t]  (two factors, the second one constant along rows)
            s = torch.randn(k, generator=g)
            a, b, s_d = A.to(device), B.to(device), s.to(device)
            keep.append(s_d)
            descs.append(ops._desc(m, n, k, ops._p(a), ops._p(b), ops._p(c), ops._lin(k), ops._lin(1), ops._lin(n), ops._lin(1), ops._lin(n), ops._lin(1),
                                   a2=ops._p(s_d), a2_r=ops._lin(0), a2_k=ops._lin(1)))
            ref = (A.double() * s.double()[None]) @ B.double()
        keep += [a, b]
        checks.append((c, ref))
    ops._gemm_group(descs)
    for i, (c, ref) in enumerate(checks):
        assert rel(c, ref) < TOL, i


@pytest.mark.parametrize("d", [3, 5, 7])
@pytest.mark.parametrize("n,mu,mv,mw", [(203, 16, 24, 8), (64, 64, 64, 64), (1000, 5, 7, 12)])
def test_gemm_repeated_two_factor_operand(device, d, n, mu, mv, mw):
    """a_rep: A[r, (u, w)] = sum_{k < d} x[k][r][u] * g[k][r][w] formed while staging (the attribute
    gradient of the self connection), against the same sum in fp64."""
    from matten_amd import ops

    gen = torch.Generator().manual_seed(d * 1000 + n)
    x = torch.randn(d, n, mu, generator=gen)
    gg = torch.randn(d, n, mw, generator=gen)
    W = torch.randn(mu, mv, mw, generator=gen)
    ref = 0.7 * torch.einsum("knu,knw,uvw->nv", x.double(), gg.double(), W.double())
    xd, gd, Wd = x.to(device), gg.to(device), W.to(device)
    c = torch.full((n, mv), float("nan"), device=device)
    ops._gemm_group([ops._desc(
        n, mv, mu * mw, ops._p(xd), ops._p(Wd), ops._p(c),
        a1_r=ops._lin(mu), a1_k=ops._imap(1, mw, 1, 0, 0),
        a2=ops._p(gd), a2_r=ops._lin(mw), a2_k=ops._imap(1, mw, 0, 0, 1),
        b_k=ops._imap(1, mw, mv * mw, 0, 1), b_j=ops._lin(mw), c_r=ops._lin(mv), c_j=ops._lin(1), alpha=0.7,
        a_rep=d, a1_rs=n * mu, a2_rs=n * mw,
    )])
    assert rel(c, ref) < TOL


def test_gemm_rejects_bad_repeat(device):
    from matten_amd import _lib, ops

    a = torch.randn(8, 8, device=device)
    c = torch.empty(8, 8, device=device)
    d = ops._desc(8, 8, 8, ops._p(a), ops._p(a), ops._p(c), ops._lin(8), ops._lin(1), ops._lin(8), ops._lin(1), ops._lin(8), ops._lin(1), a_rep=4)
    arr = (_lib.GemmDesc * 1)(d)
    assert _lib.lib().tfc_gemm_group(arr, 1, ops._stream()) < 0


@pytest.mark.parametrize(
    "hs,rows",
    [
        ([8, 64, 64, 100], 1),      # last width not a multiple of 64; a single row
        ([8, 64, 64, 100], 127),
        ([8, 33, 17, 52], 129),     # odd hidden widths, rows one past a 128-row block
        ([5, 64, 64, 64, 64, 68], 300),  # four hidden layers
        ([8, 16, 8], 1000),         # two layers
        ([8, 64, 64, 1728], 0),     # no edges
    ],
)
def test_radial_mlp_fused_edge_cases(device, hs, rows):
    """tfc_radial_mlp_fwd at the corners of what it supports (the module takes it whenever
    the hidden widths are <= 64 and the last width is a multiple of 4), forward and backward."""
    from matten_amd.nn.e3 import FullyConnectedNet
    from oracle import e3ref
    from oracle.eigenn_ref import ssp

    torch.manual_seed(50)
    o = e3ref.FullyConnectedNet(hs, act=ssp, act_key="ssp").double()
    h = FullyConnectedNet(hs, act="ssp")
    _pair_state(o, h)
    h = h.to(device)
    x = torch.randn(rows, hs[0], dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    y, yg = o(x), h(xg)
    assert yg.shape == y.shape
    if rows == 0:
        return
    assert rel(yg, y) < TOL
    go = torch.randn_like(y)
    ref = torch.autograd.grad(y, [x] + list(o.parameters()), go)
    got = torch.autograd.grad(yg, [xg] + list(h.parameters()), go.float().to(device))
    for r, g_ in zip(ref, got):
        assert rel(g_, r) < TOL


def test_radial_mlp_fwd_unsupported_shapes_are_refused(device):
    """The C entry point reports TFC_EUNSUPPORTED instead of computing something else."""
    import ctypes as C

    from matten_amd import _lib, ops

    L = _lib.lib()
    x = torch.randn(10, 8, device=device)
    w0 = torch.randn(8, 128, device=device)
    w1 = torch.randn(128, 64, device=device)
    out = torch.empty(10, 64, device=device)
    hid = torch.empty(10, 128, device=device)
    dims = (C.c_int32 * 3)(8, 128, 64)
    cw = (C.c_void_p * 2)(w0.data_ptr(), w1.data_ptr())
    ch = (C.c_void_p * 2)(hid.data_ptr(), None)
    rc = L.tfc_radial_mlp_fwd(ops._ptr(x), 10, 2, dims, cw, 1.0, ch, ops._ptr(out), ops._stream())
    assert rc < 0 and L.tfc_last_error()


def test_profile_modes(device):
    """tfc_profile_enable(2) records the convolution kernels only; (1) records the others too."""
    import ctypes as C

    from matten_amd import _lib, ops
    from matten_amd.nn.utils import UVUTensorProduct

    L = _lib.lib()
    h = UVUTensorProduct("8x0e+4x1o", "0e+1o+2e", "8x0e+4x1o", mlp_input_size=4)
    n, e = 10, 50
    ei = random_graph(n, e, 60, device)
    x = torch.randn(n, h.tp.plan.d_in, device=device)
    y = torch.randn(e, 9, device=device)
    w = torch.randn(e, h.tp.plan.w_numel, device=device)
    a = torch.randn(64, 64, device=device)
    c = torch.empty(64, 64, device=device)

    def names(mode):
        L.tfc_profile_reset()
        L.tfc_profile_enable(mode)
        try:
            ops.conv(x, y, w, h.tp.plan, ops.EdgeOrdering(ei, n), 1.0)
            ops._gemm(64, 64, 64, ops._p(a), ops._p(a), ops._p(c), ops._lin(64), ops._lin(1), ops._lin(64), ops._lin(1), ops._lin(64), ops._lin(1), tag=b"gemm_test")
        finally:
            L.tfc_profile_enable(0)
        buf = C.create_string_buffer(1 << 14)
        assert L.tfc_profile_report(buf, len(buf)) >= 0
        L.tfc_profile_reset()
        return {ln.split()[0] for ln in buf.value.decode().splitlines()}

    conv_only, everything = names(2), names(1)
    assert conv_only and all(nm.startswith("tfc_conv_") for nm in conv_only)
    assert conv_only < everything and "gemm_test" in everything


def test_linear_c_entry_points(device):
    """tfc_linear_fwd / tfc_linear_bwd called directly (as a C caller would): instructions in e3nn's order,
    two in-blocks feeding one out-block (accumulated in separate launches), an out-block nobody writes
    and an in-block nobody reads left untouched, gt handed over ready-made."""
    import ctypes as C

    from matten_amd import _lib, ops

    L = _lib.lib()
    n, d_in, d_out = 70, 8 + 4 + 3 * 5 + 6, 5 + 3 * 7 + 2
    # in: 8x0e | 4x0e | 5x1o | 6x2e(unread)    out: 5x0e | 7x1o | 2x0o(unwritten)
    spec = [(0, 8, 0, 5, 1, 0), (8, 4, 0, 5, 1, 40), (12, 5, 5, 7, 3, 60)]
    pw = [12 ** -0.5, 12 ** -0.5, 5 ** -0.5]
    ins = (_lib.LinearIns * 3)()
    for a, q, p in zip(ins, spec, pw):
        a.in_off, a.mul_in, a.out_off, a.mul_out, a.dim, a.w_off, a.path_weight = *q, p
    wn = 40 + 20 + 35
    g = torch.Generator().manual_seed(70)
    x = torch.randn(n, d_in, generator=g, dtype=torch.float64)
    w = torch.randn(wn, generator=g, dtype=torch.float64)
    go = torch.randn(n, d_out, generator=g, dtype=torch.float64)

    def ref(x, w):
        out = torch.zeros(n, d_out, dtype=torch.float64)
        for (io, mi, oo, mo, d, wo), p in zip(spec, pw):
            xb = x[:, io : io + mi * d].reshape(n, mi, d)
            W = w[wo : wo + mi * mo].reshape(mi, mo)
            out[:, oo : oo + mo * d] += p * torch.einsum("nuk,uw->nwk", xb, W).reshape(n, mo * d)
        return out

    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    o_ref = ref(xr, wr)
    gx_ref, gw_ref = torch.autograd.grad(o_ref, [xr, wr], go)
    P, st = ops._ptr, ops._stream()
    xd, wd, god = x.float().to(device), w.float().to(device), go.float().to(device)
    out = torch.full((n, d_out), 7.0, device=device)
    xt = torch.empty(n * d_in, device=device)
    _lib.check(L.tfc_linear_fwd(ins, 3, P(xd), P(wd), n, d_in, d_out, wn, P(xt), P(out), st))
    assert rel(out[:, :26], o_ref[:, :26]) < TOL and bool((out[:, 26:] == 7.0).all())
    gt = torch.empty(n * d_out, device=device)
    dx = torch.full((n, d_in), 3.0, device=device)
    dw = torch.empty(wn, device=device)
    need = int(L.tfc_linear_workspace_floats(ins, 3, n))
    ws = torch.empty(max(need, 1), device=device)
    _lib.check(L.tfc_linear_bwd(ins, 3, P(xt), P(god), P(wd), n, d_in, d_out, wn, P(gt), 0, P(dx), P(dw), P(ws), need, st))
    assert rel(dx[:, :27], gx_ref[:, :27]) < TOL and bool((dx[:, 27:] == 3.0).all())
    assert rel(dw, gw_ref) < TOL
    # gt ready-made, gout not passed; only dx wanted
    dx2 = torch.zeros(n, d_in, device=device)
    _lib.check(L.tfc_linear_bwd(ins, 3, None, None, P(wd), n, d_in, d_out, wn, P(gt), 1, P(dx2), None, None, 0, st))
    assert torch.equal(dx2[:, :27], dx[:, :27])
    # argument errors: no launch
    ins[2].out_off = 20
    assert L.tfc_linear_fwd(ins, 3, P(xd), P(wd), n, d_in, d_out, wn, P(xt), P(out), st) < 0
    ins[2].out_off = 5
    assert L.tfc_linear_bwd(ins, 3, None, P(god), P(wd), n, d_in, d_out, wn, P(gt), 0, None, P(dw), None, 0, st) < 0  # dw without xt


@pytest.mark.parametrize("dims,rows", [([8, 64, 64, 100], 333), ([6, 20, 12], 50), ([8, 40], 77)])
def test_radial_mlp_bwd_c_entry_point(device, dims, rows):
    """tfc_radial_mlp_bwd called directly against autograd on the oracle network; also with the input
    gradient and one weight gradient not requested."""
    import ctypes as C

    from matten_amd import _lib, ops
    from oracle import e3ref
    from oracle.eigenn_ref import ssp

    torch.manual_seed(71)
    o = e3ref.FullyConnectedNet(dims, act=ssp, act_key="ssp").double()
    nl = len(dims) - 1
    x = torch.randn(rows, dims[0], dtype=torch.float64, requires_grad=True)
    hs = [x]
    for layer in o:
        hs.append(layer(hs[-1]))
    go = torch.randn_like(hs[-1])
    ref = torch.autograd.grad(hs[-1], [x] + list(o.parameters()), go)
    cst = float(o[0].act.cst) if nl > 1 else 1.0
    L = _lib.lib()
    dev = lambda t: t.detach().float().to(device).contiguous()  # noqa: E731
    xd, god = dev(x), dev(go)
    wd = [dev(p) for p in o.parameters()]
    hid = [dev(h) for h in hs[1:-1]]
    vp = lambda ts: (C.c_void_p * max(len(ts), 1))(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731
    c_dims = (C.c_int32 * (nl + 1))(*dims)
    dhid = [torch.empty(rows, d, device=device) for d in dims[1:-1]]
    dx = torch.empty(rows, dims[0], device=device)
    dws = [torch.empty_like(w) for w in wd]
    need = int(L.tfc_radial_mlp_bwd_workspace_floats(rows, nl, c_dims))
    ws = torch.empty(max(need, 1), device=device)
    P = ops._ptr
    _lib.check(L.tfc_radial_mlp_bwd(P(xd), rows, nl, c_dims, vp(wd), cst, vp(hid), P(god), vp(dhid), P(dx), vp(dws), P(ws), need, ops._stream()))
    assert rel(dx, ref[0]) < TOL
    for a, b in zip(dws, ref[1:]):
        assert rel(a, b) < TOL
    dws2 = [torch.empty_like(w) for w in wd]
    dws2[0] = None
    _lib.check(L.tfc_radial_mlp_bwd(P(xd), rows, nl, c_dims, vp(wd), cst, vp(hid), P(god), vp(dhid), None, vp(dws2), None, 0, ops._stream()))
    for a, b in zip(dws2[1:], ref[2:]):
        assert rel(a, b) < TOL


def test_fctp_c_entry_points(device):
    """tfc_fctp_fwd / tfc_fctp_bwd called directly, against the oracle's FullyConnectedTensorProduct:
    forward on top of a base, then dx, dW and the per-node da with gout handed over untransposed."""
    from matten_amd import _lib, ops
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    in1, in2, out = "12x0e+8x1o+4x2e", "16x0e", "10x0o+20x0e+8x1o+4x2e"
    torch.manual_seed(80)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)  # only its instruction table is used
    spec = h.spec
    n = 90
    x = torch.randn(n, spec.d_x, dtype=torch.float64, requires_grad=True)
    a = torch.randn(n, spec.d_a, dtype=torch.float64, requires_grad=True)
    base = torch.randn(n, spec.d_out, dtype=torch.float64)
    y = base + o(x, a)
    go = torch.randn_like(y)
    gx, ga, gw = torch.autograd.grad(y, [x, a, o.weight], go)

    L = _lib.lib()
    P, st = ops._ptr, ops._stream()
    ins = ops._fctp_ins(tuple(spec.instructions))
    f = lambda t: t.detach().float().to(device).contiguous()  # noqa: E731
    xd, ad, wd, god, outd = f(x), f(a), f(o.weight), f(go), f(base)
    wn = wd.numel()
    _lib.check(L.tfc_fctp_fwd(ins, len(ins), P(xd), P(ad), P(wd), n, spec.d_x, spec.d_a, spec.d_out, wn, 1, P(outd), None, 0, st))
    assert rel(outd, y) < TOL
    gt = torch.empty(n * spec.d_out, device=device)
    dx = torch.zeros(n, spec.d_x, device=device)
    da = torch.zeros(n, spec.d_a, device=device)
    dw = torch.empty(wn, device=device)
    xt = torch.empty(n * spec.d_x, device=device)
    part = torch.empty(len(ins) * n * spec.d_a, device=device)
    need = int(L.tfc_fctp_workspace_floats(ins, len(ins), n))
    ws = torch.empty(max(need, 1), device=device)
    _lib.check(L.tfc_fctp_bwd(ins, len(ins), P(xd), P(ad), P(wd), P(god), n, spec.d_x, spec.d_a, spec.d_out, wn, P(gt), 0,
                              P(dx), P(dw), P(da), P(xt), P(part), P(ws), need, st))
    assert rel(dx, gx) < TOL and rel(da, ga) < TOL and rel(dw, gw) < TOL
    # da without its scratch is refused; a bad instruction too
    assert L.tfc_fctp_bwd(ins, len(ins), P(xd), P(ad), P(wd), P(god), n, spec.d_x, spec.d_a, spec.d_out, wn, P(gt), 0,
                          None, None, P(da), None, None, None, 0, st) < 0
    assert L.tfc_fctp_fwd(ins, len(ins), P(xd), P(ad), P(wd), n, spec.d_x, spec.d_a, spec.d_out, wn - 1, 0, P(outd), None, 0, st) < 0


# --------------------------------------------------------------------------------------- norm activation
@pytest.mark.parametrize("act", ["ssp", "silu", "sigmoid"])
@pytest.mark.parametrize("normalize", [True, False])
def test_norm_activation_vs_oracle(device, act, normalize):
    """tfc_norm_act_fwd/bwd against the oracle's NormActivation (e3nn op order), incl. all-zero copies
    (clamped at epsilon: finite output, no gradient through the norm) and l up to 3."""
    from matten_amd.nn.e3 import NormActivation
    from oracle import e3ref as E
    from oracle import eigenn_ref as R

    irreps = "5x0e+3x0o+4x1o+2x1e+3x2e+2x3o"
    eps = 1e-8 if normalize else None
    hip = NormActivation(irreps, act, normalize=normalize, epsilon=eps)
    ref = E.NormActivation(irreps, R.ACTIVATION["e"][act], normalize=normalize, epsilon=eps)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, hip.irreps_in.dim, generator=g)
    if normalize:
        x[0] = 0.0  # every copy of this row is clamped
        x[1, 8:11] = 0.0  # one 1o copy
    go = torch.randn(37, hip.irreps_in.dim, generator=g)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    (gr,) = torch.autograd.grad(yr, xr, go.double())
    xd = x.to(device).requires_grad_(True)
    yd = hip(xd)
    (gd,) = torch.autograd.grad(yd, xd, go.to(device))
    assert torch.isfinite(yd).all() and torch.isfinite(gd).all()
    # rows 0 and 1 hold clamped copies, whose scale act(1e-8) / 1e-8 is a cancellation in fp32 (ssp(1e-8)
    # rounds to 0 there, in the reference's fp32 arithmetic too): those rows are compared with the oracle
    # evaluated in fp32, the others with the fp64 value
    assert rel(yd, yr) < TOL and rel(gd[2:], gr[2:]) < TOL
    x32 = x.clone().requires_grad_(True)
    (g32,) = torch.autograd.grad(ref(x32), x32, go)
    assert rel(gd[:2], g32[:2]) < TOL
    assert str(hip.irreps_out) == str(ref.irreps_out)


@pytest.mark.parametrize("lmax", [1, 2, 3])
@pytest.mark.parametrize("normalization", ["component", "integral", "norm"])
def test_spherical_harmonics_without_normalising_the_vector(device, lmax, normalization):
    """edge_sh_normalize=False (eigenn/nn/_nequip.py:150,169): the polynomials at the edge vector itself,
    value and gradient against the oracle, through the module (positions -> EDGE_ATTRS -> d/dpos)."""
    import matten_amd.nn as H
    from matten_amd import keys as K
    from oracle import eigenn_ref as R

    g = torch.Generator().manual_seed(lmax)
    pos = torch.randn(12, 3, generator=g) * 1.5
    ei = torch.randint(0, 12, (2, 50), generator=g)
    ei = ei[:, ei[0] != ei[1]]
    go = torch.randn(ei.shape[1], (lmax + 1) ** 2, generator=g)
    outs = []
    for mod, dev, dt in ((R, "cpu", torch.float64), (H, device, torch.float32)):
        m = mod.SphericalHarmonicEdgeAttrs(lmax, edge_sh_normalization=normalization, edge_sh_normalize=False).to(dev)
        p = pos.to(dev, dt).requires_grad_(True)
        data = m({K.POSITIONS: p, K.EDGE_INDEX: ei.to(dev)})
        sh = data[K.EDGE_ATTRS]
        (gp,) = torch.autograd.grad(sh, p, go.to(dev, dt))
        outs.append((sh.detach(), gp))
    assert rel(outs[1][0], outs[0][0]) < TOL and rel(outs[1][1], outs[0][1]) < TOL


@pytest.mark.parametrize("graphs", [1, 3])
def test_cell_gradient_vs_oracle(device, graphs):
    """with_edge_vectors is differentiable in the periodic cell as well (eigenn/nn/_nequip.py:236-262):
    d/dcell and d/dpos of a scalar built from EDGE_ATTRS and EDGE_EMBEDDING, against the oracle in fp64
    (tfc_edge_cell_bwd + segment sums over each structure's nodes; single-cell and batched-cell branches)."""
    import matten_amd.nn as H
    from matten_amd import keys as K
    from matten_amd.data.crystal import collate, make_crystal
    from oracle import eigenn_ref as R

    batch = collate([make_crystal(2, 2.6, seed=70 + g) for g in range(graphs)])
    E = batch[K.EDGE_INDEX].shape[1]
    g = torch.Generator().manual_seed(graphs)
    go1, go2 = torch.randn(E, 9, generator=g), torch.randn(E, 8, generator=g)
    outs = []
    for mod, dev, dt in ((R, "cpu", torch.float64), (H, device, torch.float32)):
        sh = mod.SphericalHarmonicEdgeAttrs(2).to(dev)
        rb = mod.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0}).to(dev).to(dt)
        data = {k: (v.to(dt) if v.is_floating_point() else v).to(dev) for k, v in batch.items() if k in (K.POSITIONS, K.CELL, K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.BATCH)}
        data[K.POSITIONS].requires_grad_(True)
        data[K.CELL].requires_grad_(True)
        pos, cell = data[K.POSITIONS], data[K.CELL]
        data = rb(sh(data))
        loss = (data[K.EDGE_ATTRS] * go1.to(dev, dt)).sum() + (data[K.EDGE_EMBEDDING] * go2.to(dev, dt)).sum()
        gp, gc = torch.autograd.grad(loss, [pos, cell])
        outs.append((gp, gc))
    assert outs[1][1].shape == outs[0][1].shape == (3 * graphs, 3)
    assert rel(outs[1][0], outs[0][0]) < TOL and rel(outs[1][1], outs[0][1]) < TOL


def test_conv_fwd_with_last_mlp_layer_folded_in(device):
    """tfc_conv_fwd_fused_m (the radial network's last layer formed inside the convolution, w never written)
    against the fp64 oracle and against the shipped two-step route."""
    import ctypes as C

    from matten_amd import _lib, ops
    from matten_amd.nn.utils import UVUTensorProduct
    from oracle import eigenn_ref as R

    in1, sh, out = "40x0e+24x1o+24x1e+10x2e", "0e+1o+2e", "40x0o+40x0e+24x1o+24x1e+10x2o+10x2e"
    torch.manual_seed(4)
    o = R.UVUTensorProduct(in1, sh, out, mlp_input_size=8, mlp_hidden_size=64, mlp_num_hidden_layers=2)
    h = UVUTensorProduct(in1, sh, out, mlp_input_size=8, mlp_hidden_size=64, mlp_num_hidden_layers=2)
    n_nodes, n_edges = 6, 500
    ei = random_graph(n_nodes, n_edges, 5, "cpu")
    x = torch.randn(n_nodes, o.tp.irreps_in1.dim, dtype=torch.float64)
    y = R.spherical_harmonics(sh, torch.randn(n_edges, 3, dtype=torch.float64), True, "component")
    hid = torch.randn(n_edges, 64, dtype=torch.float64)
    w_last = torch.randn(64, o.tp.weight_numel, dtype=torch.float64)
    ref = R.scatter_add(o.tp(x[ei[0]], y, hid @ w_last / 8.0), ei[1], n_nodes) * 0.3
    xg, yg, hg, wg = (t.float().to(device) for t in (x, y, hid, w_last))
    ordering = ops.EdgeOrdering(ei.to(device), n_nodes)
    plan = h.tp.plan
    rows = ordering.by_dst
    M = ordering.m_matrices(plan, yg, rows)
    got = torch.full((n_nodes, plan.d_out), float("nan"), device=device)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.tfc_conv_fwd_fused_m(plan.handle(xg.device), P(xg), P(M), P(hg), P(wg), 64, 0.125, P(rows[0]), P(rows[1]), P(rows[2]), n_nodes, n_edges,
                                0.3, P(got), st)
    assert rc == 0, L.tfc_last_error()
    assert rel(got, ref) < TOL
    shipped = h.tp(xg, yg, (hg @ wg) * 0.125, ordering, 0.3)
    assert rel(got, shipped) < TOL
    assert L.tfc_conv_fwd_fused_m(plan.handle(xg.device), P(xg), P(M), P(hg), P(wg), 32, 0.125, P(rows[0]), P(rows[1]), P(rows[2]), n_nodes,
                                  n_edges, 0.3, P(got), st) == -2


@pytest.mark.parametrize("in1,mid", [("32x0e+16x1o+8x2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e"), ("100x0e+70x1o", "100x0e+70x1o+70x1e")])
def test_conv_channel_contiguous_copies(device, in1, mid):
    """tfc_conv_fwd_m_t / tfc_conv_bwd_m_t: the second, channel-contiguous copy of out / dx is bit-identical to
    tfc_ir_transpose of the first one, block by block, and the o3.Linear that follows picks it up
    (tfc_linear_fwd_t with xt_ready) with a bit-identical result."""
    from matten_amd import _lib, ops
    from matten_amd.nn.e3 import Linear
    from matten_amd.nn.utils import UVUTensorProduct

    torch.manual_seed(50)
    h = UVUTensorProduct(in1, "0e+1o+2e", mid, mlp_input_size=4)
    plan = h.tp.plan
    assert plan.out_blocks
    n, e = 41, 1100
    ei = random_graph(n, e, 51, device)
    o = ops.EdgeOrdering(ei, n)
    x = torch.randn(n, plan.d_in, device=device)
    y = torch.randn(e, plan.d_sh, device=device)
    w = torch.randn(e, plan.w_numel, device=device)
    L, P, st = _lib.lib(), ops._ptr, ops._stream()
    hd = plan.handle(x.device)
    M = o.m_matrices(plan, y, o.by_dst)
    out = torch.empty(n, plan.d_out, device=device)
    out_t = torch.full((n * plan.d_out,), float("nan"), device=device)
    rp, eid, nbr = o.by_dst
    _lib.check(L.tfc_conv_fwd_m_t(hd, P(x), P(M), P(w), P(rp), P(eid), P(nbr), n, e, 0.7, P(out), P(out_t), st))
    out_plain = torch.empty_like(out)
    _lib.check(L.tfc_conv_fwd_m(hd, P(x), P(M), P(w), P(rp), P(eid), P(nbr), n, e, 0.7, P(out_plain), st))
    assert torch.equal(out, out_plain)
    ops.clear_caches()
    assert torch.equal(out_t, ops._ir_transpose(out, plan.out_blocks))  # the blocks tile the row: every element is set

    go = torch.randn(n, plan.d_out, device=device)
    rp, eid, nbr = o.by_src
    dx, dw, dy = torch.empty_like(x), torch.empty_like(w), torch.empty_like(y)
    dyp = torch.empty(int(L.tfc_conv_dy_floats(hd, e)), device=device)
    dx_t = torch.full((x.numel(),), float("nan"), device=device)
    _lib.check(L.tfc_conv_bwd_m_t(hd, P(x), P(M), P(w), P(go), P(rp), P(eid), P(nbr), n, e, 0.7, P(dx), P(dx_t), P(dw), P(dy), P(dyp), st))
    dx2, dw2, dy2 = torch.empty_like(x), torch.empty_like(w), torch.empty_like(y)
    _lib.check(L.tfc_conv_bwd_m(hd, P(x), P(M), P(w), P(go), P(rp), P(eid), P(nbr), n, e, 0.7, P(dx2), P(dw2), P(dy2), P(dyp), st))
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(dy, dy2)
    ops.clear_caches()
    assert torch.equal(dx_t, ops._ir_transpose(dx, plan.in_blocks))

    # a null second copy without blocks is fine; a non-null one before tfc_conv_plan_set_out_blocks is refused
    bare = ops.ConvPlan(plan.paths, plan.d_in, plan.d_sh, plan.d_out, plan.w_numel)
    rp, eid, nbr = o.by_dst
    assert L.tfc_conv_fwd_m_t(bare.handle(x.device), P(x), P(M), P(w), P(rp), P(eid), P(nbr), n, e, 0.7, P(out), P(out_t), st) < 0
    bad = (_lib.IrBlock * 1)()
    bad[0].off, bad[0].mul, bad[0].dim = 0, plan.d_out // 5, 5  # paths with l3 = 0 do not fit a 5-wide block
    assert L.tfc_conv_plan_set_out_blocks(bare.handle(x.device), bad, 1) < 0

    # end to end: conv -> o3.Linear with and without the hand-over, forward and backward
    lin = Linear(h.irreps_out, "24x0e+12x1o+6x2e").to(device)
    res = {}
    for on in (True, False):
        ops.clear_caches()
        p2 = ops.ConvPlan(plan.paths, plan.d_in, plan.d_sh, plan.d_out, plan.w_numel, out_blocks=plan.out_blocks if on else None)
        xr = x.clone().requires_grad_(True)
        pre = Linear(in1, in1).to(device)
        torch.manual_seed(52)
        with torch.no_grad():
            pre.weight.copy_(torch.randn_like(pre.weight))
        msg = ops.conv(pre(xr), y, w, p2, o, 0.7)
        hit = ops._transposed_get(msg, ops._linear_in_blocks(lin.spec))
        assert (hit is not None) == on
        z = lin(msg)
        gz = torch.Generator().manual_seed(53)
        gx, gw1, gw2 = torch.autograd.grad(z, [xr, pre.weight, lin.weight], torch.randn(z.shape, generator=gz).to(device))
        res[on] = (z.detach(), gx, gw1, gw2)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


def test_gate_channel_contiguous_copies(device):
    """tfc_gate_fwd_t / tfc_gate_bwd_t: the second copy equals tfc_ir_transpose of the first, which equals the
    plain call's result."""
    from matten_amd import _lib, ops
    from matten_amd.nn.e3 import Gate

    gate = Gate("20x0e+12x0o", ["ssp", "tanh"], "9x0e+7x0e", ["ssp", "silu"], "9x1o+7x2e")
    spec = gate.spec
    assert spec.in_blocks and spec.out_blocks
    n = 77
    g = torch.Generator().manual_seed(60)
    x = torch.randn(n, spec.d_in, generator=g).to(device)
    go = torch.randn(n, spec.d_out, generator=g).to(device)
    L, P, st = _lib.lib(), ops._ptr, ops._stream()
    out, out2 = torch.empty(n, spec.d_out, device=device), torch.empty(n, spec.d_out, device=device)
    out_t = torch.full((n * spec.d_out,), float("nan"), device=device)
    _lib.check(L.tfc_gate_fwd_t(spec.c_array(), len(spec.segs), P(x), n, spec.d_in, spec.d_out, P(out), ops._ir_blocks(spec.out_blocks), len(spec.out_blocks), P(out_t), st))
    _lib.check(L.tfc_gate_fwd(spec.c_array(), len(spec.segs), P(x), n, spec.d_in, spec.d_out, P(out2), st))
    assert torch.equal(out, out2)
    ops.clear_caches()
    assert torch.equal(out_t, ops._ir_transpose(out, spec.out_blocks))
    dx, dx2 = torch.empty_like(x), torch.empty_like(x)
    dx_t = torch.full((x.numel(),), float("nan"), device=device)
    _lib.check(L.tfc_gate_bwd_t(spec.c_array(), len(spec.segs), P(x), P(go), n, spec.d_in, spec.d_out, P(dx), ops._ir_blocks(spec.in_blocks), len(spec.in_blocks), P(dx_t), st))
    _lib.check(L.tfc_gate_bwd(spec.c_array(), len(spec.segs), P(x), P(go), n, spec.d_in, spec.d_out, P(dx2), st))
    assert torch.equal(dx, dx2)
    ops.clear_caches()
    assert torch.equal(dx_t, ops._ir_transpose(dx, spec.in_blocks))
    # blocks that do not tile the row take the per-column kernels: the listed blocks are written, the rest left alone
    part = spec.out_blocks[1:]
    out_p = torch.full((n * spec.d_out,), 7.0, device=device)
    _lib.check(L.tfc_gate_fwd_t(spec.c_array(), len(spec.segs), P(x), n, spec.d_in, spec.d_out, P(out2), ops._ir_blocks(part), len(part), P(out_p), st))
    assert torch.equal(out, out2)
    off0, mul0, dim0 = spec.out_blocks[0]
    lo, hi = off0 * n, (off0 + mul0 * dim0) * n
    assert torch.all(out_p[lo:hi] == 7.0) and torch.equal(out_p[hi:], out_t[hi:]) and torch.equal(out_p[:lo], out_t[:lo])
    part = spec.in_blocks[:-1]
    dx_p = torch.full((x.numel(),), 7.0, device=device)
    _lib.check(L.tfc_gate_bwd_t(spec.c_array(), len(spec.segs), P(x), P(go), n, spec.d_in, spec.d_out, P(dx2), ops._ir_blocks(part), len(part), P(dx_p), st))
    offl, mull, diml = spec.in_blocks[-1]
    assert torch.equal(dx, dx2) and torch.equal(dx_p[: offl * n], dx_t[: offl * n]) and torch.all(dx_p[offl * n :] == 7.0)
    # the module route hands the copies over
    ops.clear_caches()
    xr = x.clone().requires_grad_(True)
    y = gate(xr)
    assert ops._transposed_get(y, spec.out_blocks) is not None
    (gx,) = torch.autograd.grad(y, xr, go)
    assert torch.equal(gx, dx) and ops._transposed_get(gx, spec.in_blocks) is not None
    bad = (_lib.IrBlock * 1)()
    bad[0].off, bad[0].mul, bad[0].dim = spec.d_out - 2, 2, 3
    assert L.tfc_gate_fwd_t(spec.c_array(), len(spec.segs), P(x), n, spec.d_in, spec.d_out, P(out), bad, 1, P(out_t), st) < 0


def test_fctp_class_rows_entry_point(device):
    """tfc_fctp_class_rows: da[n] = inv_count[cls[n]] * sum_i part[i, cls[n]], bit-exact against the same sum in
    torch (ascending i); classes outside [0, r) give zero rows."""
    from matten_amd import _lib, ops

    g = torch.Generator().manual_seed(80)
    ni, r, d_a, n = 3, 4, 24, 333
    part = torch.randn(ni, r, d_a, generator=g).to(device)
    inv_count = (1.0 / torch.randint(1, 90, (r,), generator=g).float()).to(device)
    cls = torch.randint(0, r, (n,), generator=g).to(device)
    cls[5] = r  # outside
    cls[6] = -1
    da = torch.full((n, d_a), float("nan"), device=device)
    _lib.check(_lib.lib().tfc_fctp_class_rows(ops._ptr(part), ni, r, d_a, ops._ptr(inv_count), ops._ptr(cls), n, ops._ptr(da), ops._stream()))
    rows = ((part[0] + part[1]) + part[2]) * inv_count[:, None]
    ref = rows[cls.clamp(0, r - 1)]
    ref[5] = 0
    ref[6] = 0
    assert torch.equal(da, ref)
    assert _lib.lib().tfc_fctp_class_rows(ops._ptr(part), 0, r, d_a, ops._ptr(inv_count), ops._ptr(cls), n, ops._ptr(da), ops._stream()) < 0


def test_class_table_matches_its_definition(device):
    """tfc_class_table (the per-batch class structure of the self connection's low-rank route): one-hot rows, first
    row and inverse size of each class, with an empty class, padding classes and a label out of range."""
    from matten_amd import ops

    g = torch.Generator().manual_seed(5)
    n, u = 1000, 6
    labels = torch.randint(0, u, (n,), generator=g)
    labels[labels == 4] = 1  # class 4 has no member
    labels[:3] = torch.tensor([5, 2, 2])
    inv, (idx, mask), Cm, u_out, r, inv_count = ops._class_structure_static(labels.to(device), u)
    assert (u_out, r) == (u, 8) and Cm.shape == (n, r)
    ref_C = torch.zeros(n, r)
    ref_C[torch.arange(n), labels] = 1.0
    assert torch.equal(Cm.cpu(), ref_C)
    counts = ref_C.sum(0)
    assert torch.equal(inv_count.cpu(), 1.0 / counts.clamp_min(1.0))
    assert torch.equal(mask.cpu().reshape(-1), (counts > 0).float())
    for s in range(r):
        want = int((labels == s).nonzero()[0]) if counts[s] > 0 else 0
        assert int(idx[s]) == want, s
    # a label outside [0, u): zero row, counted nowhere
    bad = labels.clone()
    bad[7] = 6
    bad[8] = -1
    _, (idx2, mask2), Cm2, _, _, inv2 = ops._class_structure_static(bad.to(device), u)
    assert Cm2[7].abs().sum().item() == 0 and Cm2[8].abs().sum().item() == 0
    ref2 = torch.zeros(n, r)
    ok = (bad >= 0) & (bad < u)
    ref2[torch.arange(n)[ok], bad[ok]] = 1.0
    assert torch.equal(Cm2.cpu(), ref2)
    assert torch.equal(inv2.cpu(), 1.0 / ref2.sum(0).clamp_min(1.0))
    # more classes than the kernel gathers per workgroup: the direct route
    big = torch.randint(0, 300, (5000,), generator=g)
    _, (idx3, mask3), Cm3, _, r3, inv3 = ops._class_structure_static(big.to(device), 300)
    ref3 = torch.zeros(5000, r3)
    ref3[torch.arange(5000), big] = 1.0
    assert torch.equal(Cm3.cpu(), ref3) and torch.equal(inv3.cpu(), 1.0 / ref3.sum(0).clamp_min(1.0))
    first = torch.full((r3,), 5000, dtype=torch.int64).scatter_reduce_(0, big, torch.arange(5000), reduce="amin")
    assert torch.equal(idx3.cpu(), torch.where(first < 5000, first, torch.zeros_like(first)))


def test_fctp_rows_entry_points_match_their_definition(device):
    """tfc_fctp_rows_w / _dw (weight re-associations of the low-rank self connection) through the C ABI against
    einsum in fp64, on instructions of different sizes with an attribute block that does not start at column 0."""
    import ctypes as C

    from matten_amd import _lib, ops

    g = torch.Generator().manual_seed(11)
    r, d_a = 8, 40
    # (x_off, mul_u, a_off, mul_v, out_off, mul_w, dim, w_off, alpha)
    ins, w_off = [], 0
    for mu, ao, mv, mw in ((16, 0, 24, 12), (8, 24, 16, 64), (5, 0, 40, 7)):
        ins.append((0, mu, ao, mv, 0, mw, 1, w_off, 1.0))
        w_off += mu * mv * mw
    rows = torch.randn(r, d_a, generator=g).to(device)
    weight = torch.randn(w_off, generator=g).to(device)
    L = _lib.lib()
    c_ins = ops._fctp_ins(tuple(ins))
    n_wp = int(L.tfc_fctp_rows_numel(c_ins, len(c_ins), r))
    assert n_wp == sum(mu * r * mw for _, mu, _, _, _, mw, _, _, _ in ins)
    wp = torch.full((n_wp,), float("nan"), device=device)
    _lib.check(L.tfc_fctp_rows_w(c_ins, len(c_ins), ops._ptr(rows), r, d_a, ops._ptr(weight), weight.numel(), ops._ptr(wp), ops._stream()))
    dwp = torch.randn(n_wp, generator=g).to(device)
    dw = torch.full((w_off,), float("nan"), device=device)
    _lib.check(L.tfc_fctp_rows_dw(c_ins, len(c_ins), ops._ptr(rows), r, d_a, ops._ptr(dwp), dw.numel(), ops._ptr(dw), ops._stream()))
    torch.cuda.synchronize()
    p_off = 0
    for i, (_, mu, ao, mv, _, mw, _, wo, _) in enumerate(ins):
        W = weight[wo : wo + mu * mv * mw].view(mu, mv, mw).double().cpu()
        R = rows[:, ao : ao + mv].double().cpu()
        ref_wp = torch.einsum("sv,uvw->usw", R, W)
        got_wp = wp[p_off : p_off + mu * r * mw].view(mu, r, mw).double().cpu()
        assert (got_wp - ref_wp).abs().max() <= 1e-5 * ref_wp.abs().max(), i
        G = dwp[p_off : p_off + mu * r * mw].view(mu, r, mw).double().cpu()
        ref_dw = torch.einsum("sv,usw->uvw", R, G)
        got_dw = dw[wo : wo + mu * mv * mw].view(mu, mv, mw).double().cpu()
        assert (got_dw - ref_dw).abs().max() <= 1e-5 * ref_dw.abs().max(), i
        p_off += mu * r * mw
    # limits are refused, not overrun
    many = ops._fctp_ins(tuple((0, 4, 0, 4, 0, 4, 1, 0, 1.0) for _ in range(33)))
    assert L.tfc_fctp_rows_w(many, 33, ops._ptr(rows), r, d_a, ops._ptr(weight), weight.numel(), ops._ptr(wp), ops._stream()) != 0
    assert L.tfc_fctp_rows_w(c_ins, len(c_ins), ops._ptr(rows), 6, d_a, ops._ptr(weight), weight.numel(), ops._ptr(wp), ops._stream()) != 0
