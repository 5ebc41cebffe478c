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

    assert seen == sorted(seen) and len(seen) == 14 and seen[-1] == eng.n_conv_w
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
