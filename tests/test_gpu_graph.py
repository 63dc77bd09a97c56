"""-m gpu: a whole step (edge encoders, conv layers, backward to every parameter and to the positions)
captured into a HIP graph (matten_amd.graphs.GraphedStep) replays to the same bits as the launch-by-launch
step, also after new positions and new parameters have been copied into the captured buffers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager_step(device):
    from matten_amd import keys as K
    from matten_amd.graphs import GraphedStep
    from tests.parity import CONFIGS, Stack, _prep, make_batch

    cfg = CONFIGS["minimal"]
    torch.manual_seed(35)
    batch = make_batch(cfg)
    model = Stack("hip", cfg).to(device)
    params = list(model.parameters())
    b0 = _prep(batch, str(device), torch.float32)
    pos_in = b0[K.POSITIONS].detach().clone()  # the static input buffer of the captured step
    n_graphs = cfg["graphs"]

    def step():
        for p in params:
            p.grad = None
        b = dict(b0)
        pos = pos_in.detach().requires_grad_(True)
        b[K.POSITIONS] = pos
        feats, energy, _ = model(b, n_graphs)
        energy.sum().backward()
        return energy.detach(), pos.grad, [p.grad for p in params]

    def snapshot(out):
        return out[0].clone(), out[1].clone(), [g.clone() for g in out[2]]

    for _ in range(2):
        step()  # eager steps on the default stream first, as a training script would have run
    replay = GraphedStep(step)
    for trial in range(3):
        if trial:
            with torch.no_grad():  # new inputs and new parameters, written into the buffers the graph reads
                pos_in.add_(0.01 * torch.randn(pos_in.shape, generator=torch.Generator().manual_seed(trial)).to(device))
                for p in params:
                    p.mul_(1.0 + 0.01 * trial)
        got = snapshot(replay())
        got2 = snapshot(replay())
        ref = snapshot(step())
        torch.cuda.synchronize()
        for a, b, c in zip([got[0], got[1]] + got[2], [ref[0], ref[1]] + ref[2], [got2[0], got2[1]] + got2[2]):
            assert torch.equal(a, b) and torch.equal(a, c)
        assert got[1].abs().max().item() > 0


def test_replay_takes_new_batches_by_copy_in(device):
    """rebuild_topology=True: the edge orderings and the class table are part of the graph, so a NEW batch of the same
    (padded) shape copied into the static buffers replays to the bits of the eager step on it; padding by null edges
    changes nothing (checked against the unpadded batch)."""
    from matten_amd import keys as K
    from matten_amd.data.crystal import pad_null_edges
    from matten_amd.graphs import GraphedStep
    from tests.parity import CONFIGS, Stack, _prep, make_batch

    cfg = CONFIGS["c3_lowrank"]  # >= 64 atoms: the self connection builds its class table (low-rank route)
    torch.manual_seed(35)
    raw = [make_batch(cfg, seed0=1000 + 100 * i) for i in range(3)]
    for i, drop in ((1, 3), (2, 10)):  # (equal cells give equal counts: make them differ)
        raw[i][K.EDGE_INDEX] = raw[i][K.EDGE_INDEX][:, :-drop].contiguous()
        raw[i][K.EDGE_CELL_SHIFT] = raw[i][K.EDGE_CELL_SHIFT][:-drop].contiguous()
    e_pad = max(int(b[K.EDGE_INDEX].shape[1]) for b in raw) + 7
    assert len({int(b[K.EDGE_INDEX].shape[1]) for b in raw}) == 3
    batches = [_prep(pad_null_edges(b, e_pad), str(device), torch.float32) for b in raw]
    assert len({int(b[K.POSITIONS].shape[0]) for b in batches}) == 1
    model = Stack("hip", cfg).to(device)
    params = list(model.parameters())
    n_graphs = cfg["graphs"]
    from matten_amd.graphs import PackedBatch

    packed_static = PackedBatch(batches[0])  # the buffers the graph reads: views into one flat buffer
    packed = [PackedBatch(b, like=packed_static) for b in batches]
    static = packed_static.tensors

    def run(b):
        for p in params:
            p.grad = None
        b = dict(b)
        pos = b[K.POSITIONS].detach().requires_grad_(True)
        b[K.POSITIONS] = pos
        _, energy, _ = model(b, n_graphs)
        energy.sum().backward()
        return energy.detach(), pos.grad, [p.grad for p in params]

    def snapshot(out):
        return [out[0].clone(), out[1].clone()] + [g.clone() for g in out[2]]

    # null edges are exactly null
    ref_unpadded = snapshot(run(_prep(raw[1], str(device), torch.float32)))
    ref_padded = snapshot(run(batches[1]))
    for j, (a, b) in enumerate(zip(ref_unpadded, ref_padded)):
        if j < 2:  # energy, forces: the null edges add exact zeros to the per-node sums
            assert torch.equal(a, b), j
        else:  # sums over all edges (weight gradients of the radial network) are cut into other ranges: rounding only
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-30), j

    replay = GraphedStep(lambda: run(static), rebuild_topology=True)
    for i in (1, 2, 0, 2):
        packed_static.load(packed[i])  # one copy for the whole batch
        got = snapshot(replay())
        ref = snapshot(run(batches[i]))
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b), i
        assert got[1].abs().max().item() > 0


def test_replay_refuses_a_changed_topology(device):
    """Default mode: the orderings are not in the graph, so writing to a topology tensor must not replay silently."""
    from matten_amd import keys as K
    from matten_amd.graphs import GraphedStep
    from tests.parity import CONFIGS, Stack, _prep, make_batch

    cfg = CONFIGS["smoke"]
    torch.manual_seed(35)
    b0 = _prep(make_batch(cfg), str(device), torch.float32)
    model = Stack("hip", cfg).to(device)

    def step():
        b = dict(b0)
        b[K.POSITIONS] = b0[K.POSITIONS].detach()
        with torch.no_grad():
            return model(b, cfg["graphs"])[1]

    replay = GraphedStep(step, static_topology=[b0[K.EDGE_INDEX], b0["species_index"]])
    replay()
    with torch.no_grad():
        b0[K.EDGE_INDEX].copy_(b0[K.EDGE_INDEX].flip(0))
    with pytest.raises(RuntimeError, match="topology"):
        replay()


def test_an_earlier_graph_survives_the_capture_of_a_later_one(device):
    """A GraphedStep clears the ops' caches when it is built.  Tensors an EARLIER graph reads at captured addresses (the
    side stream's workspaces, made in its warm-up) must outlive that: the earlier graph keeps them (ops.snapshot_caches)
    and replays to the bits of the eager step after two later captures and eager steps in between."""
    from matten_amd import keys as K
    from matten_amd.graphs import GraphedStep
    from tests.parity import CONFIGS, Stack, _prep, make_batch

    cfg = CONFIGS["minimal"]
    torch.manual_seed(35)
    b0 = _prep(make_batch(cfg), str(device), torch.float32)
    other = _prep(make_batch(CONFIGS["smoke"], seed0=77), str(device), torch.float32)
    model = Stack("hip", cfg).to(device)
    model2 = Stack("hip", CONFIGS["smoke"]).to(device)
    params = list(model.parameters())

    def run(m, b, n_graphs):
        for p in m.parameters():
            p.grad = None
        b = dict(b)
        pos = b[K.POSITIONS].detach().requires_grad_(True)
        b[K.POSITIONS] = pos
        _, energy, _ = m(b, n_graphs)
        energy.sum().backward()
        return energy.detach(), pos.grad, [p.grad for p in m.parameters()]

    first = GraphedStep(lambda: run(model, b0, cfg["graphs"]))
    ref = [t.clone() for t in (lambda o: [o[0], o[1]] + o[2])(first())]
    # two more captures (each drops every cache), with eager work that reuses whatever memory was handed back
    for _ in range(2):
        later = GraphedStep(lambda: run(model2, other, CONFIGS["smoke"]["graphs"]))
        later()
        run(model2, other, CONFIGS["smoke"]["graphs"])
        junk = [torch.full((1 << 22,), float("nan"), device=device) for _ in range(8)]  # noqa: F841 (fills freed blocks)
        del junk
    torch.cuda.synchronize()
    got = (lambda o: [o[0], o[1]] + o[2])(first())
    torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert params and got[1].abs().max().item() > 0


def test_rebuilt_topology_graph_on_one_stream(device, monkeypatch):
    """The step captured on ONE stream (TFC_MLP_SIDE_STREAM=0) with its topology: such a graph is a single chain of nodes, in
    which a hipMemsetAsync node was seen to replay with a garbage byte value from the second replay on (ROCm 7.2;
    tools/selftest/graph_memset_probe.py) - tfc_csr_build zeroed its counts that way and the second batch took an illegal
    memory access at config 4.  The library zeroes with a kernel now: two further batches replay to the bits of the eager step."""
    from matten_amd import keys as K
    from matten_amd import ops
    from matten_amd.data.crystal import pad_null_edges
    from matten_amd.graphs import GraphedStep, PackedBatch
    from matten_amd.nn import point_conv
    from tests.parity import CONFIGS, Stack, _prep, make_batch

    monkeypatch.setattr(ops, "MLP_SIDE_STREAM", False)
    point_conv.reset_prefetch_state()
    cfg = CONFIGS["c3_lowrank"]
    torch.manual_seed(35)
    raw = [make_batch(cfg, seed0=1000 + 100 * i) for i in range(3)]
    e_pad = max(int(b[K.EDGE_INDEX].shape[1]) for b in raw) + 5
    batches = [_prep(pad_null_edges(b, e_pad), str(device), torch.float32) for b in raw]
    model = Stack("hip", cfg).to(device)
    params = list(model.parameters())
    packed_static = PackedBatch(batches[0])
    packed = [PackedBatch(b, like=packed_static) for b in batches]
    static = packed_static.tensors

    def run(b):
        for p in params:
            p.grad = None
        b = dict(b)
        pos = b[K.POSITIONS].detach().requires_grad_(True)
        b[K.POSITIONS] = pos
        _, energy, _ = model(b, cfg["graphs"])
        energy.sum().backward()
        return [energy.detach(), pos.grad] + [p.grad for p in params]

    replay = GraphedStep(lambda: run(static), rebuild_topology=True)
    for i in (0, 0, 1, 2, 1):  # (the same batch twice first: the second replay is where a stale memset shows)
        packed_static.load(packed[i])
        got = [t.clone() for t in replay()]
        ref = run(batches[i])
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b), i
    point_conv.reset_prefetch_state()
