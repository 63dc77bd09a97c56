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
