"""-m gpu: the whole hot path (edge encoders + stacked PointConvMessagePassing layers) through
the C ABI against the CPU oracle: node features, energies, forces, every parameter gradient.
Tolerance 1e-5 relative (fp32, max-norm), as BASELINE.json's north_star states."""
import pytest
import torch

from tests.parity import CONFIGS, run_layer_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["smoke", "minimal", "lmax3", "c3", "c3_lowrank", "c3_lowrank_nohint", "norm_act", "c4"])
def test_stack_parity(device, config):
    from matten_amd import ops

    before = dict(ops.ROUTE_COUNTS)
    out = run_layer_parity(config, device=str(device), verbose=True)
    took_low_rank = ops.ROUTE_COUNTS["fctp_low_rank"] > before["fctp_low_rank"]
    assert took_low_rank == config.startswith("c3_lowrank"), "self connection took an unexpected route"
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

    def run(side, prefetch, after):
        monkeypatch.setattr(ops, "MLP_SIDE_STREAM", side)
        monkeypatch.setattr(ops, "MLP_PREFETCH", prefetch)
        monkeypatch.setattr(ops, "MLP_PREFETCH_AFTER_CONV", after)
        m = Stack("hip", cfg)
        m.load_state_dict(state)
        outs = [run_stack(m, batch, str(device), torch.float32) for _ in range(3)]
        torch.cuda.synchronize()
        for o in outs[1:]:
            for k, v in outs[0].items():
                assert (v is None and o[k] is None) or torch.equal(v, o[k]), k
        return outs[0]

    ref = run(False, False, False)
    for mode in [(True, False, False), (True, True, False), (True, True, True)]:
        got = run(*mode)
        for k, v in ref.items():
            assert (v is None and got[k] is None) or torch.equal(v, got[k]), (mode, k)
