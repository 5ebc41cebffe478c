"""Data-parallel hooks of the train step on real kernels, single process.

The collectives themselves are covered on CPU (tests/test_parallel_gloo.py, gloo, world size 2) and
by the driver's multi-GPU bench over RCCL; two ranks cannot share one GPU under RCCL, and gloo with
device tensors dead-locked intermittently on a shared GPU, so no multi-process test runs here.
What is checked: backward with the `on_ready` hook (prefix notifications, L2 term after the reduce: the path
every rank takes when world > 1) produces the gradients of the plain path, announces a growing
prefix that ends at the conv range, and a bucket list built over it covers the buffer once."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V, E, C, H, W, B = 300, 128, 18, 32, 32, 2


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    grid = torch.zeros((B, H, W), dtype=torch.int32)
    occ = torch.rand((B, H, W), generator=g) < 0.15
    grid[occ] = torch.randint(1, V, (int(occ.sum()),), generator=g, dtype=torch.int32)
    labels = torch.zeros((B, H, W), dtype=torch.int32)
    fg = occ & (torch.rand((B, H, W), generator=g) < 0.3)
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g, dtype=torch.int32)
    return grid, labels


def _engine(dtype="f32"):
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIERes(V, C, p)
    net.build_loss()
    return Engine(net, "cuda:0", dtype=dtype, seed=0)


def test_backward_hook_path_matches_plain_path():
    from cutie_amd.parallel import GradBuckets
    eng = _engine()
    grid, labels = _inputs(40)
    P = eng.forward(grid, labels=labels)
    eng.loss_backward(P)
    torch.cuda.synchronize()
    plain = eng.grads_flat.cpu().numpy().copy()
    reg_plain = eng.losses(P)[1]
    # run-to-run noise of the plain path itself: the instance-norm moments are float atomics, the
    # last-bit differences grow layer by layer and a ReLU on its kink flips a whole local gradient
    nrm = np.linalg.norm(plain.astype(np.float64))
    noise = 0.0
    for _ in range(3):          # the largest of three samples: one sample alone can come out small
        P = eng.forward(grid, labels=labels)
        eng.loss_backward(P)
        torch.cuda.synchronize()
        noise = max(noise, np.linalg.norm((eng.grads_flat.cpu().numpy() - plain).astype(np.float64)))
    print("run-to-run gradient noise (rel-L2): %.2e" % (noise / nrm))

    buckets = GradBuckets(eng.grads_flat, eng.n_conv_w, n_buckets=3)
    launched, seen = [], []
    buckets._launch = lambda lo, hi: launched.append((lo, hi))       # record instead of all-reducing
    def hook(upto):
        seen.append(upto)
        buckets.ready(upto)
    P = eng.forward(grid, labels=labels)
    eng.loss_backward(P, on_ready=hook)
    buckets._launch(buckets.next_lo, eng.grads_flat.numel())         # what finish() sends last
    eng.add_l2_after_reduce(P, 1)                                    # the L2 term follows the (here: absent) reduce
    torch.cuda.synchronize()
    hooked = eng.grads_flat.cpu().numpy()

    # the weight gradients of the data-parallel step go out as two batched calls (the front half of the conv range while
    # backward still runs, the rest at its end); layer by layer (CUTIE_WGRAD_BATCH=0) there is one notification per layer
    assert seen == sorted(seen) and len(seen) == (2 if eng.wgrad_batch else 14) and seen[-1] == eng.n_conv_w
    assert seen[0] >= eng.n_conv_w // 2
    assert launched[0][0] == 0 and launched[-1][1] == eng.grads_flat.numel()
    assert all(a[1] == b[0] for a, b in zip(launched, launched[1:]))  # contiguous, no overlap
    assert len(launched) == 4
    diff = np.linalg.norm((hooked - plain).astype(np.float64))
    print("hook path vs plain path (rel-L2): %.2e" % (diff / nrm))
    assert diff <= 4 * noise + 1e-5 * nrm and diff < 2e-2 * nrm
    assert abs(eng.losses(P)[1] - reg_plain) < 1e-5 * abs(reg_plain)


def test_forward_is_bit_reproducible_and_backward_nearly():
    """No float atomics on the instance-norm path (per-tile / per-slice slots added in slot order): two
    forward passes give bit-identical logits and ReLU patterns; what is left run to run is the f32
    scatter-add of the embedding gradient, at rounding level."""
    eng = _engine("bf16")
    grid, labels = _inputs(42)
    P = eng.forward(grid, labels=labels)
    logits = eng.get(P, "cls_logits").clone()
    a8 = eng.relu_mask(P, "encoder1_8").clone()
    eng.loss_backward(P)
    g1 = eng.grads_flat.clone()
    for _ in range(3):
        P = eng.forward(grid, labels=labels)
        assert torch.equal(eng.get(P, "cls_logits"), logits) and torch.equal(eng.relu_mask(P, "encoder1_8"), a8)
        eng.loss_backward(P)
        dense = slice(0, eng.n_dense)
        assert torch.equal(eng.grads_flat[dense], g1[dense])              # conv kernels and biases: bit-identical
        emb = (eng.grads_flat[eng.n_dense:] - g1[eng.n_dense:]).double().norm() / g1[eng.n_dense:].double().norm()
        assert float(emb) < 1e-5


def test_trainer_single_rank_has_no_buckets():
    from cutie_amd.train import Trainer
    eng = _engine()
    tr = Trainer(eng, learning_rate=1e-3)
    assert tr.world == 1 and tr.buckets is None
    grid, labels = _inputs(41)
    out = tr.step({"grid_table": grid, "gt_classes": labels})
    assert np.isfinite(out["total_loss"]) and out["global_norm"] > 0


class _Done(object):
    def wait(self):
        return True


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_world_size_two_emulated_on_one_gpu(monkeypatch, dtype):
    """The arithmetic of a two-rank step on real kernels without a second GPU.  dist.all_reduce is replaced by a stub
    that either records a rank's buffers or adds the other rank's recorded ones, so that two Trainers (rank 0 and rank 1,
    one shard of the batch each) go through the path every rank takes at world > 1 - backward with the bucket hook, the
    embedding-rows norm in the last collective, the L2 term after the reduce, Adam with grad_scale 1 / world.  At
    each of three steps the two replicas must agree to rounding, and must equal Adam on the mean of the two shards'
    plain-path gradients (computed by another engine with no hook and no collective, started from the replicas' state) -
    three steps, because the first Adam step is +-lr whatever the gradient's scale."""
    import torch.distributed as dist
    from cutie_amd import ops, parallel, train
    from cutie_amd.train import Trainer

    grid, labels = _inputs(43)
    g2, l2 = _inputs(44)
    grid, labels = torch.cat([grid, g2]), torch.cat([labels, l2])            # 4 documents: ranks take 0::2 and 1::2
    shards = [(grid[r::2].contiguous(), labels[r::2].contiguous()) for r in range(2)]
    mode = {"what": None, "store": None, "k": 0}

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        if mode["what"] == "record":
            mode["store"].append(t.clone())
        else:
            t.add_(mode["store"][mode["k"]])
            mode["k"] += 1
        return _Done()

    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    engs = [_engine(dtype) for _ in range(3)]                                  # rank 0, rank 1, scratch
    ref = _engine(dtype)
    trs = []
    for r, e in enumerate(engs):
        monkeypatch.setattr(train, "dist_info", lambda r=r: (min(r, 1), 2))
        trs.append(Trainer(e, learning_rate=1e-3))
        assert trs[-1].world == 2 and trs[-1].buckets is not None
    m_ref, v_ref = torch.zeros_like(ref.params_flat), torch.zeros_like(ref.params_flat)
    start = ref.params_flat.clone()
    for step in range(3):
        # what each rank would send: run its shard on the scratch engine from the replicas' common state
        sent = []
        for r in range(2):
            s, ts = engs[2], trs[2]
            s.params_flat.copy_(engs[0].params_flat)
            s.repack()
            ts.m.copy_(trs[0].m); ts.v.copy_(trs[0].v); ts.t, ts.iter = trs[0].t, trs[0].iter
            if s.loss_scale is not None:
                s.loss_scale.copy_(engs[0].loss_scale)
            mode.update(what="record", store=[], k=0)
            ts.step({"grid_table": shards[r][0], "gt_classes": shards[r][1]}, fetch=False)
            sent.append(mode["store"])
        assert len(sent[0]) == len(sent[1]) >= 3                               # conv buckets, the rest, the embedding-rows norm
        for r in range(2):
            mode.update(what="apply", store=sent[1 - r], k=0)
            out = trs[r].step({"grid_table": shards[r][0], "gt_classes": shards[r][1]})
            assert mode["k"] == len(sent[1 - r]) and np.isfinite(out["total_loss"])
        # the same step by hand: plain-path gradients of the two shards (each with its own wd * w), summed; Adam at 1/2
        gsum, emb_sq = torch.zeros_like(ref.grads_flat), 0.0
        for r in range(2):
            P = ref.forward(shards[r][0], labels=shards[r][1])
            ref.loss_backward(P)
            gsum += ref.grads_flat
            emb_sq += float(ref.embedding_rows_sumsq(P).item())
        gn = (gsum[:ref.n_dense].double() ** 2).sum() + emb_sq
        gn = torch.tensor([float(gn)], dtype=torch.float32, device=gsum.device)
        t = step + 1
        lr_t = 1e-3 * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
        ops.adam_step(ref.params_flat, gsum, m_ref, v_ref, lr_t, 0.9, 0.999, 1e-8, gn, 10.0, grad_scale=0.5, t=t,
                      loss_scale=ref.loss_scale)
        ref.repack()
        # Adam turns a gradient component at rounding level (|g| ~ eps = 1e-8) into a step of up to lr in either direction,
        # so single elements may differ by ~lr between two summation orders: compare the movement as a whole, and
        # element-wise where the gradient is well determined
        mv_dp, mv_ref = (engs[0].params_flat - start).double(), (ref.params_flat - start).double()
        rel = float((mv_dp - mv_ref).norm() / mv_ref.norm())
        solid = gsum.abs() > 1e-4 * float(gsum.abs().max())
        worst = float((mv_dp - mv_ref)[solid].abs().max())
        print("step %d: movement rel-L2 %.2e, worst well-determined element %.2e (lr 1e-3), %d of %d elements solid, global norm %.4f"
              % (step, rel, worst, int(solid.sum()), solid.numel(), out["global_norm"]))
        print("   norms: replica %.6f by hand %.6f (dense %.6f, embedding rows %.6f)" % (
            out["global_norm"], float(gn.sqrt()) / 2, float((gsum[:ref.n_dense].double() ** 2).sum().sqrt()) / 2, emb_sq ** 0.5 / 2))
        # a + b == b + a, so real replicas stay bit-identical; here a rank's own buffer and the copy of it the other rank
        # receives come from two RUNS of the same shard, and the embedding gradient's f32 atomics are reproducible only to
        # rounding (test_forward_is_bit_reproducible_and_backward_nearly) - which Adam may turn into a step of ~lr where
        # the gradient itself is at rounding level.  So: equal where the gradient is well determined, close as a whole
        d01 = (engs[0].params_flat - engs[1].params_flat).double()
        assert float(d01[solid].abs().max()) < (5e-7 if dtype == "f32" else 5e-5), step
        assert float(d01.norm() / mv_ref.norm()) < 2e-2, step
        assert rel < (1e-3 if dtype == "f32" else 2e-2), step
        assert worst < (2e-5 if dtype == "f32" else 3e-4), step
        assert abs(out["global_norm"] - float(gn.sqrt()) / 2) < 1e-3 * out["global_norm"]
        # the next step starts from the replicas' state on both sides (m and v are then non-trivial): each step checks the
        # arithmetic of ONE step; left to themselves two trajectories drift apart through ReLU kinks and the hard-negative
        # selection (2.6e-3 of the movement after two steps), which says nothing about the collectives
        engs[1].params_flat.copy_(engs[0].params_flat)
        engs[1].repack()
        trs[1].m.copy_(trs[0].m)
        trs[1].v.copy_(trs[0].v)
        ref.params_flat.copy_(engs[0].params_flat)
        ref.repack()
        m_ref.copy_(trs[0].m)
        v_ref.copy_(trs[0].v)
        if ref.loss_scale is not None:
            ref.loss_scale.copy_(engs[0].loss_scale)
        start = ref.params_flat.clone()
