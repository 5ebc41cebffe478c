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


# ------------------------------------------------------------------------------ Trainer.step control flow, world 2
# The kernels are stubbed by recorded gradients (no GPU here); what runs is the real Trainer.step and GradBuckets:
# loss_backward(on_ready) -> finish(extra) -> add_l2_after_reduce -> global_norm_sq -> adam_step -> repack.
N_CONV, N_BIAS, N_EMB = 4096, 64, 320
WD = 5e-4


class _StubPlan(object):
    probs = None


class _StubEngine(object):
    """The Engine surface Trainer.step touches, on host tensors.  grads_of(step) supplies this replica's gradient."""

    def __init__(self, grads_of, emb_rows_sq_of):
        g = torch.Generator().manual_seed(7)
        self.n_conv_w, self.n_dense = N_CONV, N_CONV + N_BIAS
        self.params_flat = torch.randn(N_CONV + N_BIAS + N_EMB, generator=g)
        self.grads_flat = torch.zeros_like(self.params_flat)
        self.loss_scale = None
        self.grads_of, self.emb_rows_sq_of = grads_of, emb_rows_sq_of
        self.step_no, self.repacks, self.l2_pending = 0, 0, False

    def forward(self, grid, image=None, ps_1d_indices=None, labels=None):
        return _StubPlan()

    def loss_backward(self, P, on_ready=None):
        src = self.grads_of(self.step_no)
        self.grads_flat.zero_()
        lo = 0
        for hi in (700, 701, 2048, 3000, N_CONV):        # backward finishes the conv range front to back, unevenly
            self.grads_flat[lo:hi] = src[lo:hi]
            if on_ready is not None:
                on_ready(hi)
            lo = hi
        self.grads_flat[N_CONV:] = src[N_CONV:]
        self.l2_pending = on_ready is not None
        if on_ready is None:
            self.grads_flat[:N_CONV] += WD * self.params_flat[:N_CONV]
        return P

    def embedding_rows_sumsq(self, P):
        return torch.tensor([self.emb_rows_sq_of(self.step_no)], dtype=torch.float32)

    def add_l2_after_reduce(self, P, world=1):
        assert self.l2_pending
        self.grads_flat[:N_CONV] += world * WD * self.params_flat[:N_CONV]
        self.l2_pending = False

    def global_norm_sq(self, P, tf_indexed_slices=True, embedding_rows_sq=None):
        e = embedding_rows_sq if embedding_rows_sq is not None else self.embedding_rows_sumsq(P)
        return (self.grads_flat[:self.n_dense] ** 2).sum().reshape(1) + e

    def repack(self):
        self.repacks += 1
        self.step_no += 1

    def losses(self, P):
        return 0.0, 0.0, 0.0

    def get(self, P, name):
        return None


class _StubOps(object):
    """cutie_adam_step's arithmetic (csrc/optim.hip) in torch"""

    @staticmethod
    def adam_step(p, g, m, v, lr_t, b1, b2, eps, gnorm_sq, clip, grad_scale=1.0, t=0, loss_scale=None):
        norm = torch.sqrt(gnorm_sq[0]) * grad_scale
        scale = grad_scale * (clip / max(float(norm), clip))
        gi = g * scale
        m.mul_(b1).add_((1 - b1) * gi)
        v.mul_(b2).add_((1 - b2) * gi * gi)
        p.sub_(lr_t * m / (v.sqrt() + eps))

    @staticmethod
    def loss_scale_update(*a, **k):
        raise AssertionError("no loss scale in this test")


def _recorded(rank, step):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(N_CONV + N_BIAS + N_EMB, generator=g) * (30.0 if step == 1 else 0.5)     # step 1 is clipped


def _emb_sq(rank, step):
    return float(3.0 + rank + 10 * step)


def _trainer_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cutie_amd.train as train
        train.ops = _StubOps
        eng = _StubEngine(lambda s: _recorded(rank, s), lambda s: _emb_sq(rank, s))
        tr = train.Trainer(eng, learning_rate=1e-2)
        assert tr.world == world and tr.buckets is not None
        for step in range(3):
            out = tr.step({"grid_table": None, "gt_classes": None})
            assert abs(out["global_norm"] ** 2 * world ** 2 - float(eng.global_norm_sq(None, embedding_rows_sq=torch.tensor(
                [sum(_emb_sq(r, step) for r in range(world))]))[0])) < 1e-2 * out["global_norm"] ** 2 * world ** 2
            np.save(os.path.join(out_dir, "p_r%d_s%d.npy" % (rank, step)), eng.params_flat.numpy())
        assert eng.repacks == 3 and not eng.l2_pending
    finally:
        dist.destroy_process_group()


def test_trainer_step_world2_replicas_identical_and_equal_to_mean_gradient(tmp_path):
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # one process, fed the MEAN gradient of the two replicas (and the global batch's embedding rows, scaled alike)
    import cutie_amd.train as train
    saved = train.ops
    train.ops = _StubOps
    try:
        eng = _StubEngine(lambda s: sum(_recorded(r, s) for r in range(world)) / world,
                          lambda s: sum(_emb_sq(r, s) for r in range(world)) / world ** 2)
        tr = train.Trainer(eng, learning_rate=1e-2)
        assert tr.buckets is None
        for step in range(3):
            tr.step({"grid_table": None, "gt_classes": None})
            got = [np.load(tmp_path / ("p_r%d_s%d.npy" % (r, step))) for r in range(world)]
            assert np.array_equal(got[0], got[1]), "replicas diverged at step %d" % step
            np.testing.assert_allclose(got[0], eng.params_flat.numpy(), rtol=0, atol=2e-6)
    finally:
        train.ops = saved
