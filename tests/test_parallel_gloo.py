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


def _worker_extra(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, n_front = 900, 640
        store = torch.zeros(n + 64)
        flat, extra = store[:n], store[n:n + 1]          # Engine.grads_flat / Engine.grads_extra
        buckets = GradBuckets(flat, n_front, bounds=[384, n_front])
        assert buckets.bounds == [384, 640]
        for step in range(2):
            flat.copy_(_rank_grads(rank, n) * (step + 1))
            extra.fill_(float(rank + 1 + step))
            buckets.ready(384)
            buckets.ready(n_front)
            launched = len(buckets.works)
            buckets.finish(extra=extra)                   # the last slice reaches over `extra`: no collective of its own
            assert launched == 2
            np.save(os.path.join(out_dir, "x%d_s%d.npy" % (rank, step)), store.numpy())
    finally:
        dist.destroy_process_group()


def test_clip_norm_share_rides_in_the_last_slice(tmp_path):
    world = 2
    mp.spawn(_worker_extra, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for step in range(2):
        got = [np.load(tmp_path / ("x%d_s%d.npy" % (r, step))) for r in range(world)]
        assert np.array_equal(got[0], got[1])
        want = sum(_rank_grads(r, 900) for r in range(world)).numpy() * (step + 1)
        np.testing.assert_allclose(got[0][:900], want, rtol=0, atol=1e-6)
        assert got[0][900] == sum(r + 1 + step for r in range(world))
        assert not got[0][901:].any()


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
# loss_backward(on_ready, defer_l2) -> finish(extra) -> apply_update (L2 term, clip norm, Adam, repack).
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
        self.step_no, self.updates, self.l2_pending = 0, 0, False

    def forward(self, grid, image=None, ps_1d_indices=None, labels=None):
        return _StubPlan()

    def loss_backward(self, P, on_ready=None, defer_l2=None):
        assert defer_l2, "Trainer.step leaves the L2 term to apply_update()"
        src = self.grads_of(self.step_no)
        self.grads_flat.zero_()
        lo = 0
        for hi in (700, 701, 2048, 3000, N_CONV):        # backward finishes the conv range front to back, unevenly
            self.grads_flat[lo:hi] = src[lo:hi]
            if on_ready is not None:
                on_ready(hi)
            lo = hi
        self.grads_flat[N_CONV:] = src[N_CONV:]
        self.l2_pending = True
        return P

    def embedding_rows_sumsq(self, P):
        return torch.tensor([self.emb_rows_sq_of(self.step_no)], dtype=torch.float32)

    def apply_update(self, P, m, v, lr_t, b1, b2, eps, clip, world=1, t=0, emb_rows_sq=None, scale_growth_interval=1000):
        """cutie_opt_norm + cutie_opt_step's arithmetic (csrc/optim.hip) in torch: the buffer holds the SUM of the ranks'
        raw gradients, the L2 term goes in as world * wd * w, the optimizer applies 1 / world"""
        assert self.l2_pending
        e = emb_rows_sq if emb_rows_sq is not None else self.embedding_rows_sumsq(P)
        g = self.grads_flat.clone()
        g[:N_CONV] += world * WD * self.params_flat[:N_CONV]
        gnorm_sq = (g[:self.n_dense] ** 2).sum().reshape(1) + e
        grad_scale = 1.0 / world
        norm = torch.sqrt(gnorm_sq[0]) * grad_scale
        scale = grad_scale * (clip / max(float(norm), clip))
        gi = g * scale
        m.mul_(b1).add_((1 - b1) * gi)
        v.mul_(b2).add_((1 - b2) * gi * gi)
        self.params_flat.sub_(lr_t * m / (v.sqrt() + eps))
        self.l2_pending = False
        self.updates += 1
        self.step_no += 1
        self.last_gnorm_sq = gnorm_sq
        return gnorm_sq

    def losses(self, P):
        return 0.0, 0.0, 0.0

    def get(self, P, name):
        return None


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
        eng = _StubEngine(lambda s: _recorded(rank, s), lambda s: _emb_sq(rank, s))
        tr = train.Trainer(eng, learning_rate=1e-2)
        assert tr.world == world and tr.buckets is not None
        for step in range(3):
            p_before = eng.params_flat.clone()
            out = tr.step({"grid_table": None, "gt_classes": None})
            # the norm the clip saw: dense part of the reduced buffer (+ world * wd * w) and the embedding rows of BOTH ranks
            g = sum(_recorded(r, step) for r in range(world))
            want = float(((g[:N_CONV] + world * WD * p_before[:N_CONV]) ** 2).sum() + (g[N_CONV:N_CONV + N_BIAS] ** 2).sum()) \
                + sum(_emb_sq(r, step) for r in range(world))
            assert abs(out["global_norm"] ** 2 * world ** 2 - want) < 1e-4 * want
            np.save(os.path.join(out_dir, "p_r%d_s%d.npy" % (rank, step)), eng.params_flat.numpy())
        assert eng.updates == 3 and not eng.l2_pending
    finally:
        dist.destroy_process_group()


def test_trainer_step_world2_replicas_identical_and_equal_to_mean_gradient(tmp_path):
    world = 2
    mp.spawn(_trainer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    # one process, fed the MEAN gradient of the two replicas (and the global batch's embedding rows, scaled alike)
    import cutie_amd.train as train
    if True:
        eng = _StubEngine(lambda s: sum(_recorded(r, s) for r in range(world)) / world,
                          lambda s: sum(_emb_sq(r, s) for r in range(world)) / world ** 2)
        tr = train.Trainer(eng, learning_rate=1e-2)
        assert tr.buckets is None
        for step in range(3):
            tr.step({"grid_table": None, "gt_classes": None})
            got = [np.load(tmp_path / ("p_r%d_s%d.npy" % (r, step))) for r in range(world)]
            assert np.array_equal(got[0], got[1]), "replicas diverged at step %d" % step
            np.testing.assert_allclose(got[0], eng.params_flat.numpy(), rtol=0, atol=2e-6)
