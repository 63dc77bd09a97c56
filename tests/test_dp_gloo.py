"""CPU, world_size 2, gloo: the data-parallel path (matten_amd/dp.py) - graphs sharded by
rank, one flat gradient bucket, one all-reduce.  The averaged gradient must equal the
single-process gradient of the mean loss over the union of the shards."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matten_amd.dp import FlatGradBucket, shard_graphs


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))


def _data():
    g = torch.Generator().manual_seed(4)
    return torch.randn(8, 5, 6, generator=g), torch.randn(8, generator=g)  # 8 "graphs" of 5 "atoms"


def _loss(model, x, y, ids):
    e = model(x[ids]).sum(dim=(1, 2))
    return ((e - y[ids]) ** 2).mean()


def _worker(rank, world, port, out, bucket_bytes=8 << 20, overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        bucket = FlatGradBucket(model.parameters(), bucket_bytes=bucket_bytes, overlap=overlap)
        x, y = _data()
        for _ in range(2):  # second pass checks that zero() really resets the aliased grads
            bucket.zero()
            _loss(model, x, y, shard_graphs(8, rank, world)).backward()
            early = sum(w is not None for w in bucket._works)  # collectives already in flight when backward returns
            bucket.all_reduce_mean()
        if rank == 0:
            # (in parameter order, whatever the layout of the buffer)
            torch.save({"grads": torch.cat([p.grad.reshape(-1) for p in model.parameters()]), "slices": len(bucket.bounds),
                        "launched_early": early}, out)
    finally:
        dist.destroy_process_group()


def test_shard_graphs_partition():
    parts = [shard_graphs(10, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10)) and parts[1] == [1, 5, 9]


def test_flat_bucket_aliases_grads():
    model = _model()
    b = FlatGradBucket(model.parameters(), alias=True)
    x, y = _data()
    _loss(model, x, y, list(range(8))).backward()
    off = 0
    for p in reversed(list(model.parameters())):  # reverse registration order = order of readiness in backward
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off
        off += p.numel()
    assert b.bounds == [(0, off)]
    small = FlatGradBucket(_model().parameters(), bucket_bytes=64, overlap=True)  # 17 + 16 floats | 96 floats
    assert small.bounds == [(0, 17), (17, 33), (33, 129)] and small._n_in_slice == [2, 1, 1]
    assert b.flat.abs().sum() > 0
    b.all_reduce_mean()  # no process group: a no-op


@pytest.mark.parametrize("bucket_bytes,overlap,n_slices", [(8 << 20, True, 1), (64, True, 3), (64, False, 1)])
def test_two_rank_gloo_matches_single_process(tmp_path, bucket_bytes, overlap, n_slices):
    out = str(tmp_path / "flat.pt")
    port = 29500 + (os.getpid() % 2000) + n_slices + int(overlap) * 3
    mp.spawn(_worker, args=(2, port, out, bucket_bytes, overlap), nprocs=2, join=True)
    saved = torch.load(out, weights_only=True)
    got = saved["grads"]
    assert saved["slices"] == n_slices and saved["launched_early"] == (n_slices if overlap else 0)
    model = _model()
    x, y = _data()
    ref = 0.5 * (_loss(model, x, y, shard_graphs(8, 0, 2)) + _loss(model, x, y, shard_graphs(8, 1, 2)))
    grads = torch.autograd.grad(ref, list(model.parameters()))
    ref_flat = torch.cat([g.reshape(-1) for g in grads])
    assert torch.allclose(got, ref_flat, atol=1e-6)


def test_flat_bucket_survives_zero_grad_and_refuses_stale_buffers():
    """optimizer.zero_grad() (set_to_none) breaks the aliasing of p.grad to the flat buffer: zero() restores it,
    and all_reduce_mean() refuses to reduce a buffer the gradients no longer live in."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 500))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model = _model()
        b = FlatGradBucket(model.parameters(), force=True, alias=True)  # force: run the collective in a group of one
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        x, y = _data()
        b.zero()
        _loss(model, x, y, list(range(8))).backward()
        b.all_reduce_mean()
        ref = b.flat.clone()
        opt.zero_grad()  # grads -> None
        b.zero()  # re-aliased
        _loss(model, x, y, list(range(8))).backward()
        b.all_reduce_mean()
        assert torch.equal(b.flat, ref) and all(b._aliased(p) for p in model.parameters())
        opt.zero_grad()
        _loss(model, x, y, list(range(8))).backward()  # fresh .grad tensors, the buffer is stale
        with pytest.raises(RuntimeError, match="no longer aliases"):
            b.all_reduce_mean()
        # the default (alias=False): gradients stay autograd's own until the collective packs them
        model2 = _model()
        b2 = FlatGradBucket(model2.parameters(), force=True)
        for _ in range(2):
            b2.zero()
            assert all(p.grad is None for p in model2.parameters())
            _loss(model2, x, y, list(range(8))).backward()
            b2.all_reduce_mean()
            assert torch.equal(b2.flat, ref) and all(b2._aliased(p) for p in model2.parameters())
    finally:
        dist.destroy_process_group()


def _layer_worker(rank, world, port, out):
    """The conv-layer stack itself (the CPU oracle's modules: the product layers need a GPU) under the same
    data-parallel recipe bench.py uses: graphs g = rank (mod world), flat bucket, one all-reduce."""
    from matten_amd import keys as K
    from matten_amd.data.crystal import collate
    from tests.parity import CONFIGS, Stack, make_batch, _prep

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        graphs, model = _layer_setup()
        mine = [graphs[g] for g in shard_graphs(len(graphs), rank, world)]
        bucket = FlatGradBucket(model.parameters())
        bucket.zero()
        _layer_loss(model, mine).backward()
        bucket.all_reduce_mean()
        if rank == 0:
            torch.save(torch.cat([p.grad.reshape(-1) for p in model.parameters()]), out)
    finally:
        dist.destroy_process_group()


def _layer_setup():
    from matten_amd.data.crystal import synthetic_batch
    from tests.parity import CONFIGS, Stack

    cfg = dict(CONFIGS["smoke"], graphs=4)
    _, graphs = synthetic_batch(cfg["n"], cfg["a"], cfg["graphs"], seed0=1000, r_cut=5.0)
    torch.manual_seed(35)
    return graphs, Stack("oracle", cfg).double()


def _layer_loss(model, graphs):
    from matten_amd.data.crystal import collate
    from tests.parity import _prep

    b = _prep(collate(graphs), "cpu", torch.float64)
    _, energy, _ = model(b, len(graphs))
    return energy.square().mean()


def test_two_rank_gloo_conv_stack_matches_single_process(tmp_path):
    out = str(tmp_path / "layer.pt")
    port = 27500 + (os.getpid() % 2000)
    mp.spawn(_layer_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    graphs, model = _layer_setup()
    ref = 0.5 * (_layer_loss(model, [graphs[g] for g in shard_graphs(4, 0, 2)]) + _layer_loss(model, [graphs[g] for g in shard_graphs(4, 1, 2)]))
    grads = torch.autograd.grad(ref, list(model.parameters()))
    ref_flat = torch.cat([g.reshape(-1) for g in grads])
    assert ref_flat.abs().max() > 0 and torch.allclose(got, ref_flat, rtol=1e-10, atol=1e-12 * ref_flat.abs().max().item())
