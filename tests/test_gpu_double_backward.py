"""-m gpu: second-order differentiation (training on forces).  A loss on
F = -dE/dpos needs the derivative of every backward kernel; the reference gets it from autograd's double
backward through e3nn / torch (eigenn/nn/_nequip.py:231-235 keeps the edge vectors differentiable for this).
Every op's backward is itself a differentiable op here (matten_amd/ops.py: _ConvBwdFn, _LinearBwdFn,
_FctpBwdFn, _GateBwdFn, _EdgeBwdFn, the radial network's composite), running on the same HIP kernels plus
tfc_gate_bwd2 / tfc_edge_encode_bwd2.  Checked against the fp64 CPU oracle, where torch's own double
backward does the work: per op on random cotangents, and end to end as d(sum |F|^2 + sum E^2)/d(theta, pos).
Tolerance 1e-5 of the tensor's max-norm."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    s = ref.abs().max().item()
    return (a - ref).abs().max().item() / (s if s > 0 else 1.0)


def second_order(fn, inputs, seed):
    """out = fn(*inputs); first-order gradients g_i = d<out, go>/d input_i with create_graph; then the gradient of
    sum_i <g_i, r_i> with respect to every input and to go.  Random go, r_i from `seed` (fp64 on the CPU, cast
    to the inputs' dtype and device), so that oracle and product see the same cotangents."""
    out = fn(*inputs)
    gen = torch.Generator().manual_seed(seed)
    cast = lambda t: t.to(device=out.device, dtype=out.dtype)  # noqa: E731
    go = cast(torch.randn(out.shape, generator=gen, dtype=torch.float64)).requires_grad_(True)
    g1 = torch.autograd.grad(out, inputs, go, create_graph=True)
    rs = [cast(torch.randn(g.shape, generator=gen, dtype=torch.float64)) for g in g1]
    loss = sum((g * r).sum() for g, r in zip(g1, rs))
    g2 = torch.autograd.grad(loss, list(inputs) + [go], allow_unused=True)
    return [out] + list(g1) + list(g2)


def compare(ref, got, names):
    assert len(ref) == len(got)
    for i, (r, g) in enumerate(zip(ref, got)):
        if r is None or g is None:
            assert r is None or r.abs().max().item() == 0.0 or g is not None, f"{names}: output {i} missing"
            continue
        assert rel(g, r) < TOL, f"{names}: output {i} off by {rel(g, r):.2e}"


def _pair_state(oracle_mod, hip_mod):
    hip_mod.load_state_dict({k: v.float() for k, v in oracle_mod.state_dict().items()}, strict=True)


@pytest.mark.parametrize("irreps_in,irreps_out", [("32x0e+16x1o+8x2e", "24x0e+16x1o+8x2e"), ("8x0e+8x0e+3x1o", "5x0e+2x1o+2x1o+3x2e")])
def test_linear_second_order(device, irreps_in, irreps_out):
    from matten_amd.nn.e3 import Linear
    from oracle import e3ref

    torch.manual_seed(0)
    o = e3ref.Linear(irreps_in, irreps_out).double()
    h = Linear(irreps_in, irreps_out)
    _pair_state(o, h)
    h = h.to(device)
    x = torch.randn(77, o.irreps_in.dim, dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    ref = second_order(lambda a, w: torch.func.functional_call(o, {"weight": w}, (a,)), [x, o.weight], 1)
    got = second_order(lambda a, w: torch.func.functional_call(h, {"weight": w}, (a,)), [xg, h.weight], 1)
    compare(ref, got, "linear")


def test_fctp_second_order(device):
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    in1, in2, out = "16x0e+8x1o+8x2e", "12x0e", "16x0o+40x0e+8x1o+8x1e+8x2o+8x2e"
    torch.manual_seed(1)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    _pair_state(o, h)
    h = h.to(device)
    n = 41
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    a = torch.randn(n, o.irreps_in2.dim, dtype=torch.float64, requires_grad=True)
    xg, ag = (t.detach().float().to(device).requires_grad_(True) for t in (x, a))
    ref = second_order(lambda x_, a_, w: torch.func.functional_call(o, {"weight": w}, (x_, a_)), [x, a, o.weight], 2)
    got = second_order(lambda x_, a_, w: torch.func.functional_call(h, {"weight": w}, (x_, a_)), [xg, ag, h.weight], 2)
    compare(ref, got, "fctp")


def test_fctp_low_rank_forward_second_order(device):
    """The forward takes the low-rank route (species-like attributes, >= 64 nodes); the differentiable
    backward is the dense one - same sums."""
    from matten_amd import ops
    from matten_amd.nn.e3 import FullyConnectedTensorProduct
    from oracle import e3ref

    in1, in2, out = "16x0e+8x1o", "32x0e", "24x0e+8x1o"
    torch.manual_seed(11)
    o = e3ref.FullyConnectedTensorProduct(in1, in2, out).double()
    h = FullyConnectedTensorProduct(in1, in2, out)
    _pair_state(o, h)
    h = h.to(device)
    n = 150
    table = torch.randn(3, 32, dtype=torch.float64, requires_grad=True)
    idx = torch.randint(0, 3, (n,))
    x = torch.randn(n, o.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    tg, xg = (t.detach().float().to(device).requires_grad_(True) for t in (table, x))
    before = ops.ROUTE_COUNTS["fctp_low_rank"]
    ref = second_order(lambda x_, t_, w: torch.func.functional_call(o, {"weight": w}, (x_, t_[idx])), [x, table, o.weight], 3)
    got = second_order(lambda x_, t_, w: torch.func.functional_call(h, {"weight": w}, (x_, t_[idx.to(device)])), [xg, tg, h.weight], 3)
    assert ops.ROUTE_COUNTS["fctp_low_rank"] > before
    compare(ref, got, "fctp low rank")


@pytest.mark.parametrize(
    "scalars,acts_s,gates,acts_g,gated",
    [
        ("32x0o+32x0e", ["tanh", "ssp"], "48x0e", ["ssp"], "16x1o+16x1e+8x2o+8x2e"),
        ("5x0e", ["silu"], "", [], ""),
        ("3x0e", ["sigmoid"], "4x0o", ["tanh"], "4x1e"),
    ],
)
def test_gate_second_order(device, scalars, acts_s, gates, acts_g, gated):
    from matten_amd.nn.e3 import Gate
    from oracle import e3ref
    from oracle.eigenn_ref import ACTIVATION

    fn = {**ACTIVATION["e"], **ACTIVATION["o"]}
    o = e3ref.Gate(scalars, [fn[a] for a in acts_s], gates, [fn[a] for a in acts_g], gated, keys_scalars=acts_s, keys_gates=acts_g)
    h = Gate(scalars, acts_s, gates, acts_g, gated)
    torch.manual_seed(3)
    x = torch.randn(93, o.irreps_in.dim, dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    compare(second_order(o, [x], 4), second_order(h, [xg], 4), "gate")


@pytest.mark.parametrize("hs", [[8, 64, 64, 96], [4, 8, 24]])
def test_radial_mlp_second_order(device, hs):
    from matten_amd.nn.e3 import FullyConnectedNet
    from oracle import e3ref
    from oracle.eigenn_ref import ssp

    torch.manual_seed(2)
    o = e3ref.FullyConnectedNet(hs, act=ssp, act_key="ssp").double()
    h = FullyConnectedNet(hs, act="ssp")
    _pair_state(o, h)
    h = h.to(device)
    x = torch.randn(333, hs[0], dtype=torch.float64, requires_grad=True)
    xg = x.detach().float().to(device).requires_grad_(True)
    names = [n for n, _ in o.named_parameters()]

    def call(mod):
        return lambda x_, *ws: torch.func.functional_call(mod, dict(zip(names, ws)), (x_,))

    ref = second_order(call(o), [x] + list(o.parameters()), 5)
    got = second_order(call(h), [xg] + list(h.parameters()), 5)
    compare(ref, got, "radial mlp")


@pytest.mark.parametrize(
    "in1,sh,out,n_nodes,n_edges",
    [
        ("8x0e", "0e+1o+2e", "8x0o+8x0e+4x1o+4x1e+2x2o+2x2e", 10, 120),
        ("32x0e+16x1o+16x1e+8x2o+8x2e", "0e+1o+2e", "32x0o+32x0e+16x1o+16x1e+8x2o+8x2e", 30, 700),
        ("16x0e+16x1o+8x2e+4x3o", "0e+1o+2e+3o", "16x0o+16x0e+16x1o+16x1e+8x2o+8x2e+4x3o+4x3e", 12, 200),
    ],
)
def test_conv_second_order(device, in1, sh, out, n_nodes, n_edges):
    from matten_amd import ops
    from matten_amd.nn.utils import UVUTensorProduct
    from oracle import eigenn_ref as R

    torch.manual_seed(4)
    o = R.UVUTensorProduct(in1, sh, out, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1)
    h = UVUTensorProduct(in1, sh, out, mlp_input_size=4, mlp_hidden_size=8, mlp_num_hidden_layers=1)
    ei = torch.randint(0, n_nodes, (2, n_edges), generator=torch.Generator().manual_seed(5))
    x = torch.randn(n_nodes, o.tp.irreps_in1.dim, dtype=torch.float64, requires_grad=True)
    y = R.spherical_harmonics(sh, torch.randn(n_edges, 3, dtype=torch.float64), True, "component").requires_grad_(True)
    w = torch.randn(n_edges, o.tp.weight_numel, dtype=torch.float64, requires_grad=True)
    xg, yg, wg = (t.detach().float().to(device).requires_grad_(True) for t in (x, y, w))
    ordering = ops.EdgeOrdering(ei.to(device), n_nodes)
    ref = second_order(lambda x_, y_, w_: R.scatter_add(o.tp(x_[ei[0]], y_, w_), ei[1], n_nodes) * 0.41, [x, y, w], 6)
    got = second_order(lambda x_, y_, w_: h.tp(x_, y_, w_, ordering, 0.41), [xg, yg, wg], 6)
    compare(ref, got, "conv")


@pytest.mark.parametrize("lmax", [1, 2, 3])
@pytest.mark.parametrize("batched_cell", [True, False])
def test_edge_encoders_second_order(device, lmax, batched_cell):
    """pos, bessel_weights -> (vectors, lengths, spherical harmonics, radial embedding) and back, twice."""
    import matten_amd.nn as H
    from matten_amd import keys as K
    from matten_amd.data.crystal import synthetic_batch
    from oracle import eigenn_ref as R

    batch, _ = synthetic_batch(2, 2.6, 2 if batched_cell else 1, seed0=77)

    def run(M, dev, dtype):
        torch.manual_seed(9)
        base = {k: (v.to(dtype) if v.is_floating_point() else v).to(dev) for k, v in batch.items()}
        shm = M.SphericalHarmonicEdgeAttrs(irreps_edge_sh=lmax)
        rb = M.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0}).to(dev, dtype)
        pos = base[K.POSITIONS].clone().requires_grad_(True)

        def f(pos_, bw_):
            data = dict(base)
            data[K.POSITIONS] = pos_
            data = torch.func.functional_call(rb, {"basis.bessel_weights": bw_}, (shm(data),))
            return torch.cat([data[K.EDGE_ATTRS], data[K.EDGE_EMBEDDING], data[K.EDGE_VECTORS], data[K.EDGE_LENGTH][:, None]], dim=1)

        return second_order(f, [pos, rb.basis.bessel_weights], 10)

    compare(run(R, "cpu", torch.float64), run(H, device, torch.float32), f"edge encoders lmax={lmax}")


def _force_loss_grads(model, batch, device, dtype, n_graphs):
    from matten_amd import keys as K
    from tests.parity import _prep

    model = model.to(device=device, dtype=dtype)
    b = _prep(batch, device, dtype)
    _, energy, _ = model(b, n_graphs)
    (dpos,) = torch.autograd.grad(energy.sum(), [b[K.POSITIONS]], create_graph=True)
    loss = (dpos**2).sum() + (energy**2).sum()  # |F|^2 + E^2: forces and energies in one objective
    params = list(model.parameters())
    grads = torch.autograd.grad(loss, params + [b[K.POSITIONS]], allow_unused=True)
    out = {"loss": loss.detach().reshape(1), "forces": -dpos.detach()}
    for (name, _), g in zip(list(model.named_parameters()) + [("pos", None)], grads):
        out["grad:" + name] = None if g is None else g.detach()
    return out


@pytest.mark.parametrize("config", ["smoke", "minimal", "lmax3"])
def test_force_training_gradients_vs_oracle(device, config):
    """d(sum |F|^2 + sum E^2)/d(every parameter, positions) through the whole stack: embedding, edge encoders, conv
    layers (radial networks, tensor products, linears, self connections, gates)."""
    from tests.parity import CONFIGS, Stack, make_batch

    cfg = CONFIGS[config]
    torch.manual_seed(35)
    batch = make_batch(cfg)
    oracle = Stack("oracle", cfg)
    state = {k: v.clone() for k, v in oracle.state_dict().items()}
    ref = _force_loss_grads(oracle, batch, "cpu", torch.float64, cfg["graphs"])
    prod = Stack("hip", cfg)
    prod.load_state_dict(state, strict=True)
    got = _force_loss_grads(prod, batch, str(device), torch.float32, cfg["graphs"])
    torch.cuda.synchronize()
    worst = {}
    for k, r in ref.items():
        g = got[k]
        if r is None:
            assert g is None or g.abs().max().item() == 0.0, k
            continue
        assert g is not None, f"{k}: no gradient from the product"
        worst[k] = rel(g, r)
    bad = {k: v for k, v in worst.items() if v > TOL}
    assert not bad, f"{config}: {bad}"


def test_force_training_step_through_the_energy_model(device):
    """The reference-shaped model (model_factory.create_energy_model: NodewiseLinear tail, NodewiseReduce) takes an
    optimiser step on a force-matching loss; the loss goes down and nothing in the graph is silently constant."""
    from matten_amd import model_factory
    from matten_amd.data.crystal import synthetic_batch
    from tests import models as M

    batch, _ = synthetic_batch(2, 2.6, 2, seed0=500, r_cut=5.0)
    torch.manual_seed(35)
    hp = dict(M.MINIMAL)
    model = model_factory.create_energy_model(hp, {"allowed_species": (1, 6, 8, 14), "average_num_neighbors": 26.0}).to(device)
    b0 = {k: v.to(device) for k, v in batch.items()}
    target = torch.randn(b0["pos"].shape, generator=torch.Generator().manual_seed(1)).to(device) * 0.1
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        b = dict(b0)
        b["pos"] = b0["pos"].clone().requires_grad_(True)
        e = model(b)["total_energy"]
        (dpos,) = torch.autograd.grad(e.sum(), [b["pos"]], create_graph=True)
        loss = ((-dpos - target) ** 2).mean()
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        losses.append(loss.item())
        opt.step()
    assert losses[-1] < losses[0]


def test_third_order_raises(device):
    """What is not built says so instead of returning a silently constant gradient."""
    from matten_amd.nn.e3 import Gate

    h = Gate("4x0e", ["ssp"], "2x0e", ["ssp"], "2x1o")
    x = torch.randn(5, h.irreps_in.dim, device=device, requires_grad=True)
    (g1,) = torch.autograd.grad(h(x).sum(), [x], create_graph=True)
    (g2,) = torch.autograd.grad(g1.square().sum(), [x], create_graph=True)
    with pytest.raises(RuntimeError):  # (torch: "once_differentiable" or "not used in the graph", by version)
        torch.autograd.grad(g2.square().sum(), [x])


@pytest.mark.parametrize("normalize,act", [(True, "ssp"), (True, "tanh"), (False, "silu")])
def test_norm_activation_second_order_vs_oracle(device, normalize, act):
    """activation_type="norm" (eigenn/nn/utils.py:142-150) under create_graph=True: the gradient of a function of the
    first gradient, with respect to the input and to the upstream gradient (tfc_norm_act_bwd2), one clamped copy included."""
    from matten_amd.nn.e3 import NormActivation
    from oracle import e3ref
    from oracle.eigenn_ref import ACTIVATION

    fn = {**ACTIVATION["e"], **ACTIVATION["o"]}[act]
    irreps = "6x0e+5x1o+4x2e+3x3o"
    o = e3ref.NormActivation(irreps, fn, normalize=normalize)
    h = NormActivation(irreps, act, normalize=normalize)
    torch.manual_seed(12)
    x = torch.randn(41, o.irreps_in.dim, dtype=torch.float64)
    if act == "tanh":
        # a copy below epsilon: constant scale act(eps) / eps, no gradient through its norm (tanh: that ratio is 1 in
        # fp32 as well; ssp(1e-8) is below fp32's resolution next to ln 2, in any fp32 implementation)
        x[3, 6:9] = 0.0
    w1, w2 = torch.randn_like(x), torch.randn_like(x)
    res = []
    for mod, dev, dt in ((o, "cpu", torch.float64), (h, device, torch.float32)):
        xx = x.to(dev, dt).requires_grad_(True)
        gg = w1.to(dev, dt).requires_grad_(True)
        (g1,) = torch.autograd.grad(mod(xx), [xx], gg, create_graph=True)
        loss = (g1 * w2.to(dev, dt)).sum() + g1.square().sum()
        res.append(torch.autograd.grad(loss, [xx, gg]))
    for a, b in zip(res[1], res[0]):
        assert rel(a, b) < TOL


@pytest.mark.parametrize("graphs", [1, 3])
def test_cell_gradient_second_order_vs_oracle(device, graphs):
    """The gradient with respect to the periodic cell is itself differentiable (eigenn/nn/_nequip.py:236-262 under
    create_graph=True): d/d(pos, cell) of a function of dE/dcell and dE/dpos, single-cell and batched-cell branches."""
    import matten_amd.nn as H
    from matten_amd import keys as K
    from matten_amd.data.crystal import collate, make_crystal
    from oracle import eigenn_ref as R

    batch = collate([make_crystal(2, 2.6, seed=80 + g) for g in range(graphs)])
    E = batch[K.EDGE_INDEX].shape[1]
    g = torch.Generator().manual_seed(graphs)
    go1, go2 = torch.randn(E, 9, generator=g), torch.randn(E, 8, generator=g)
    wc, wp = torch.randn(3 * graphs, 3, generator=g), torch.randn(batch[K.POSITIONS].shape, generator=g)
    outs = []
    for mod, dev, dt in ((R, "cpu", torch.float64), (H, device, torch.float32)):
        sh = mod.SphericalHarmonicEdgeAttrs(2).to(dev)
        rb = mod.RadialBasisEdgeEncoding(basis_kwargs={"num_basis": 8, "r_max": 5.0}, cutoff_kwargs={"r_max": 5.0}).to(dev).to(dt)
        data = {k: (v.to(dt) if v.is_floating_point() else v).to(dev) for k, v in batch.items() if k in (K.POSITIONS, K.CELL, K.EDGE_INDEX, K.EDGE_CELL_SHIFT, K.BATCH)}
        data[K.POSITIONS].requires_grad_(True)
        data[K.CELL].requires_grad_(True)
        pos, cell = data[K.POSITIONS], data[K.CELL]
        data = rb(sh(data))
        e = (data[K.EDGE_ATTRS] * go1.to(dev, dt)).sum() + (data[K.EDGE_EMBEDDING] * go2.to(dev, dt)).sum()
        gp, gc = torch.autograd.grad(e, [pos, cell], create_graph=True)
        loss = (gc * wc.to(dev, dt)).sum() + 0.5 * gc.square().sum() + (gp * wp.to(dev, dt)).sum()
        outs.append(torch.autograd.grad(loss, [pos, cell]))
    for a, b in zip(outs[1], outs[0]):
        assert rel(a, b) < TOL
