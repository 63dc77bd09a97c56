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
    b = FlatGradBucket(model.parameters())
    x, y = _data()
    _loss(model, x, y, list(range(8))).backward()
    off = 0
    for p in reversed(list(model.parameters())):  # reverse registration order = order of readiness in backward
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off
        off += p.numel()
    assert b.bounds == [(0, off)]
    small = FlatGradBucket(_model().parameters(), bucket_bytes=64)  # 17 + 16 floats | 96 floats
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
