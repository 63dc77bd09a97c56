"""-m gpu: the whole hot path (edge encoders + stacked PointConvMessagePassing layers) through
the C ABI against the CPU oracle: node features, energies, forces, every parameter gradient.
Tolerance 1e-5 relative (fp32, max-norm), as BASELINE.json's north_star states."""
import pytest
import torch

from tests.parity import CONFIGS, run_layer_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["smoke", "minimal", "lmax3", "c3", "c3_lowrank", "c3_lowrank_nohint", "norm_act", "c4", "c5_layer0", "c4_l2", "c3_cell", "c4_cell64"])
def test_stack_parity(device, config):
    from matten_amd import ops

    before = dict(ops.ROUTE_COUNTS)
    out = run_layer_parity(config, device=str(device), verbose=True)
    took_low_rank = ops.ROUTE_COUNTS["fctp_low_rank"] > before["fctp_low_rank"]
    assert took_low_rank == CONFIGS[config].get("low_rank", config.startswith("c3_lowrank")), "self connection took an unexpected route"
    bad = {k: v for k, v in out["report"].items() if v[0] > out["tol"]}
    assert out["ok"], f"worst {out['worst']:.2e} > {out['tol']:.0e}: {bad}"


def test_native_library_is_loaded(device):
    """The product path has no fallback: the HIP library must be the thing that ran."""
    from matten_amd import _lib

    assert _lib.lib().tfc_version() >= 100
    with open("/proc/self/maps") as f:
        assert "libtfc.so" in f.read()


def test_cpu_tensors_are_refused(device):
    from matten_amd.nn.e3 import Linear

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Linear("4x0e", "4x0e")(torch.randn(3, 4))


def test_stream_modes_give_identical_bits(device, monkeypatch):
    """The side stream and the prefetch of the next layer's radial network only reorder independent work:
    energies, forces and every parameter gradient are bit-identical with one stream, with the side stream
    but no prefetch, with the prefetch (default), and with the prefetch queued behind the convolution - and
    over repeated passes (the prefetch only engages from the second pass on)."""
    from matten_amd import ops
    from tests.parity import CONFIGS, Stack, make_batch, run_stack

    cfg = CONFIGS["minimal"]
    torch.manual_seed(35)
    batch = make_batch(cfg)
    model = Stack("hip", cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run(side, prefetch, after, early=True):
        monkeypatch.setattr(ops, "MLP_SIDE_STREAM", side)
        monkeypatch.setattr(ops, "MLP_PREFETCH", prefetch)
        monkeypatch.setattr(ops, "MLP_PREFETCH_AFTER_CONV", after)
        monkeypatch.setattr(ops, "MLP_PREFETCH_RECORD_EARLY", early)
        m = Stack("hip", cfg)
        m.load_state_dict(state)
        outs = [run_stack(m, batch, str(device), torch.float32) for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs[1:]:
            for k, v in outs[0].items():
                assert (v is None and o[k] is None) or torch.equal(v, o[k]), k
        return outs[0]

    ref = run(False, False, False)
    for mode in [(True, False, False), (True, True, False), (True, True, True), (True, True, False, False), (True, True, True, False)]:
        got = run(*mode)
        for k, v in ref.items():
            assert (v is None and got[k] is None) or torch.equal(v, got[k]), (mode, k)


def _conv_pass(model, data0, emb):
    """The conv layers of a ConvStack on a data dict that carries a GIVEN edge-embedding tensor."""
    from matten_amd import keys as K

    data = dict(data0)
    data[K.EDGE_EMBEDDING] = emb
    for layer in model.convs:
        data = layer(data)
    return data[K.NODE_FEATURES]


def _encoded(model, batch, device):
    """Embedding + edge encoders once; returns a dict the conv layers can run on repeatedly."""
    from matten_amd import keys as K
    from tests.parity import _prep

    b = _prep(batch, device, torch.float32)
    with torch.no_grad():
        w = model.embedding.weight
        x = torch.nn.functional.one_hot(b["species_index"], w.shape[0]).to(w.dtype) @ w
        b[K.NODE_FEATURES] = b[K.NODE_ATTRS] = x
        b[K.NODE_ATTRS_CLASSES] = b["species_index"]
        b = model.radial_basis(model.spharm_edges(b))
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in b.items()}


def test_prefetch_never_serves_weights_of_old_parameters(device, monkeypatch):
    """Passes that reuse ONE edge-embedding tensor, with an in-place parameter update in between (an
    optimiser step) and a load_state_dict: the prefetching path must give the bits of the path without
    prefetch, and the radial network must run exactly once per layer and pass (no prefetch across the
    end of a pass)."""
    from matten_amd import _lib, ops
    from matten_amd import keys as K
    from matten_amd.nn import point_conv
    from tests.parity import CONFIGS, Stack, make_batch

    cfg = CONFIGS["minimal"]
    torch.manual_seed(35)
    batch = make_batch(cfg)
    ref_model = Stack("hip", cfg).to(device)
    state = {k: v.clone() for k, v in ref_model.state_dict().items()}
    data0 = _encoded(ref_model, batch, str(device))
    emb = data0[K.EDGE_EMBEDDING]

    def run(prefetch, early=True):
        monkeypatch.setattr(ops, "MLP_SIDE_STREAM", True)
        monkeypatch.setattr(ops, "MLP_PREFETCH", prefetch)
        monkeypatch.setattr(ops, "MLP_PREFETCH_RECORD_EARLY", early)
        point_conv.reset_prefetch_state()
        m = Stack("hip", cfg).to(device)
        m.load_state_dict(state)
        radial = [p for layer in m.convs for p in layer.message_fn.tp.weight_nn.parameters()]
        outs = []
        L = _lib.lib()
        L.tfc_profile_reset()
        L.tfc_profile_enable(1)
        for step in range(5):
            out = _conv_pass(m, data0, emb)
            grads = torch.autograd.grad(out.square().sum(), radial)
            outs.append([out.detach().clone()] + [g.clone() for g in grads])
            with torch.no_grad():
                if step == 3:
                    m.load_state_dict(state)  # copies in place: same addresses, new versions
                else:
                    for p, g in zip(radial, grads):
                        p.add_(g, alpha=-1e-3 / (1.0 + g.abs().max().item()))
        torch.cuda.synchronize()
        L.tfc_profile_enable(0)
        launches = _lib.profile_report()["mlp_fused_fwd"][0]
        L.tfc_profile_reset()
        return outs, launches

    ref, n_ref = run(False)
    assert n_ref == 5 * len(ref_model.convs)
    for early in (True, False):
        got, n_got = run(True, early)
        assert n_got == n_ref, f"{n_got} radial-network launches with prefetch, {n_ref} without"
        for a, b in zip(ref, got):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
    assert not torch.equal(ref[0][0], ref[1][0])  # the parameter update did change the result


def test_two_models_interleaved_on_one_embedding(device, monkeypatch):
    """Two models whose layers alternate on the same embedding tensor: whatever the prefetch heuristic
    learns from that order, every result equals the un-prefetched one."""
    from matten_amd import ops
    from matten_amd import keys as K
    from matten_amd.nn import point_conv
    from tests.parity import CONFIGS, Stack, make_batch

    cfg = CONFIGS["minimal"]
    torch.manual_seed(35)
    batch = make_batch(cfg)
    models = [Stack("hip", cfg).to(device) for _ in range(2)]
    data0 = _encoded(models[0], batch, str(device))
    emb = data0[K.EDGE_EMBEDDING]

    def interleaved():
        datas = [dict(data0, **{K.EDGE_EMBEDDING: emb}) for _ in models]
        for i in range(len(models[0].convs)):
            for j, m in enumerate(models):
                datas[j] = m.convs[i](datas[j])
        return [d[K.NODE_FEATURES].detach().clone() for d in datas]

    monkeypatch.setattr(ops, "MLP_SIDE_STREAM", True)
    monkeypatch.setattr(ops, "MLP_PREFETCH", False)
    point_conv.reset_prefetch_state()
    ref = interleaved()
    monkeypatch.setattr(ops, "MLP_PREFETCH", True)
    for _ in range(3):
        for a, b in zip(ref, interleaved()):
            assert torch.equal(a, b)
