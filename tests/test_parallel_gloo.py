"""Data-parallel plumbing on CPU: two processes, gloo backend, 127.0.0.1.

What runs here is exactly what runs over RCCL on the GPU box - GradBuckets only calls
torch.distributed.all_reduce on slices of a flat buffer - minus the kernels that fill it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cutie_amd.parallel import GradBuckets, dist_info, shard_documents


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_grads(rank, n):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(n, generator=g)


def _worker(rank, world, port, n_total, n_front, pieces, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dist_info() == (rank, world)
        flat = torch.zeros(n_total)
        buckets = GradBuckets(flat, n_front, n_buckets=4)
        src = _rank_grads(rank, n_total)
        for step in range(2):               # the object is reused step after step
            flat.zero_()
            # "backward": the buffer becomes final front to back, in uneven pieces
            lo = 0
            for hi in pieces:
                flat[lo:hi] = src[lo:hi] * (step + 1)
                buckets.ready(hi)
                lo = hi
            flat[lo:] = src[lo:] * (step + 1)      # biases + embedding come last
            buckets.finish()
            np.save(os.path.join(out_dir, "r%d_s%d.npy" % (rank, step)), flat.numpy())
        # documents: every rank takes its stride of the global batch
        docs = list(range(10))
        mine = shard_documents(docs, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        assert sorted(sum(gathered, [])) == docs
        assert all(len(set(a) & set(b)) == 0 for i, a in enumerate(gathered) for b in gathered[i + 1:])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_front,pieces", [(1000, [130, 131, 640, 1000]), (4096, [4096]), (64, [10, 64])])
def test_grad_buckets_world2(tmp_path, n_front, pieces):
    world, n_total = 2, n_front + 300
    mp.spawn(_worker, args=(world, _free_port(), n_total, n_front, pieces, str(tmp_path)), nprocs=world, join=True)
    want = sum(_rank_grads(r, n_total) for r in range(world)).numpy()
    for step in range(2):
        got = [np.load(tmp_path / ("r%d_s%d.npy" % (r, step))) for r in range(world)]
        assert np.array_equal(got[0], got[1])                    # replicas stay identical
        np.testing.assert_allclose(got[0], want * (step + 1), rtol=0, atol=1e-6)


def test_bucket_bounds_cover_the_front_range_once():
    flat = torch.zeros(5000)
    b = GradBuckets(flat, 4100, n_buckets=4)
    assert b.bounds[-1] == 4100 and b.bounds == sorted(set(b.bounds))
    assert all(x % 64 == 0 for x in b.bounds[:-1])
    assert len(b.bounds) <= 5
    b1 = GradBuckets(flat, 10, n_buckets=4)
    assert b1.bounds == [10]


def test_shard_documents_single_process():
    assert dist_info() == (0, 1)
    assert shard_documents(list("abcdefg"), 1, 3) == ["b", "e"]
    assert shard_documents([], 0, 2) == []
