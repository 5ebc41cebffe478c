"""Two checks the round-2 review asked for:

1. north_star's argmax criterion (per-cell agreement >= 99.9 %) measured over ALL cells on a network with real margins.  A
   randomly initialised CUTIE puts most cells within a hair of a tie (instance norm over nearly constant logits), which says
   nothing about a trained model; so CUTIE-ASPP is first TRAINED here - on the device, f32 path, a learnable synthetic task
   (the label of a cell is a function of its token id) - and the trained weights are then run through the f64 oracle, the
   f32 path and the f16 headline path.
2. gradient parity of the f32 path against the oracle's OWN ReLU pattern (no device masks handed to the oracle), on a case
   built so that no pre-activation of any layer lies within 1e-6 of zero (seed found by search: min |z| = 8.6e-6 over 201,856
   pre-activations) - a wrong pattern carried inside y could not pass this one.
The oracle is PARITY UNPINNED against TensorFlow (oracle/cutie_oracle.py)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cutie_oracle as co  # noqa: E402

DEV = "cuda:0"
V, E, C = 600, 128, 18


def _ns(B):
    return types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)


def _learnable(B, H, W, seed):
    """12 % of the cells hold a token; a token whose id is a multiple of 3 is a field of class 1 + id % (C - 1)"""
    g = torch.Generator().manual_seed(seed)
    grid = torch.zeros((B, H, W), dtype=torch.int64)
    occ = torch.rand((B, H, W), generator=g) < 0.12
    ids = torch.randint(2, V, (B, H, W), generator=g)
    grid[occ] = ids[occ]
    labels = torch.where((grid > 0) & (grid % 3 == 0), 1 + grid % (C - 1), torch.zeros_like(grid))
    return grid, labels


def test_argmax_agreement_on_a_trained_network_over_all_cells():
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    B, H, W = 4, 32, 32
    net = CUTIERes(V, C, _ns(B))
    net.build_loss()
    eng = Engine(net, DEV, dtype="f32", seed=0)
    tr = Trainer(eng, learning_rate=1e-3)
    first = last = None
    STEPS = 1200
    for k in range(STEPS):
        grid, labels = _learnable(B, H, W, 1000 + k % 8)          # eight batches, round robin
        out = tr.step({"grid_table": grid.unsqueeze(-1), "gt_classes": labels}, fetch=(k in (0, STEPS - 1)))
        if k == 0:
            first = out["model_loss"]
        if k == STEPS - 1:
            last = out["model_loss"]
    print("trained %d steps in f32: model loss %.4f -> %.4f" % (STEPS, first, last))
    assert last < 0.7 * first
    params = {k: torch.from_numpy(v).double() for k, v in eng.state_dict().items()}
    grid, labels = _learnable(B, H, W, 2000)                      # a batch the training never saw
    L = co.aspp_forward(params, grid)
    ref = L["cls_logits"].reshape(-1, C)
    ref_arg = ref.argmax(-1)
    top2 = ref.topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1]) / float(ref.abs().max())
    acc = float((ref_arg == labels.reshape(-1)).double().mean())
    print("oracle: accuracy on the unseen batch %.4f; top-2 margin / max |logit|: median %.3f, share below 2e-2: %.5f"
          % (acc, float(margin.median()), float((margin < 2e-2).double().mean())))
    for dtype, tol in (("f32", 1e-4), ("f16", 2e-2)):
        e = Engine(net, DEV, dtype=dtype)
        e.load_params(params)
        P = e.forward(grid.unsqueeze(-1))
        got = e.get(P, "cls_logits")[..., :C].double().cpu().reshape(-1, C)
        err = float((got - ref).norm() / ref.norm())
        agree = float((got.argmax(-1) == ref_arg).double().mean())
        print("%s: logits rel-L2 %.2e, argmax agreement over ALL %d cells %.5f" % (dtype, err, ref.shape[0], agree))
        assert err < tol, (dtype, err)
        assert agree >= 0.999, (dtype, agree)


def test_f32_gradients_against_the_oracles_own_relu_pattern():
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    B, H, W, seed = 1, 8, 8, 25
    g = torch.Generator().manual_seed(seed)
    grid = torch.zeros((B, H, W), dtype=torch.int64)
    occ = torch.rand((B, H, W), generator=g) < 0.3
    ids = torch.randint(2, V, (B, H, W), generator=g)
    grid[occ] = ids[occ]
    labels = torch.zeros((B, H, W), dtype=torch.int64)
    fg = occ & (torch.rand((B, H, W), generator=g) < 0.4)
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g)
    params = co.init_params(V, E, C, seed=0)
    o = co.loss_and_grads(params, grid, labels)                   # no masks, no cell weights: the oracle by itself
    pre = {}
    co.aspp_forward(params, grid, pre=pre)
    zmin = min(float(z.abs().min()) for z in pre.values())
    assert zmin > 1e-6, zmin                                       # the case is built so (header)
    net = CUTIERes(V, C, _ns(B))
    net.build_loss()
    eng = Engine(net, DEV, dtype="f32")
    eng.load_params(params)
    P = eng.forward(grid.unsqueeze(-1), labels=labels)
    eng.loss_backward(P)
    flips = 0
    for k, z in pre.items():
        flips += int(((z > 0) != eng.relu_mask(P, k).cpu()).sum())
    print("min |pre-activation| %.2e; ReLU decisions that differ from the oracle's: %d" % (zmin, flips))
    assert flips == 0
    _, w_ref, _, _ = co.hnm_loss(co.softmax_ce(o["layers"]["cls_logits"].reshape(-1, C), labels.reshape(-1)).detach(), labels, 3)
    assert torch.equal(P.weight.cpu() != 0, w_ref != 0)            # the same hard negatives
    rel = lambda got, ref: float((got.detach().double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    worst = {}
    for k, gref in o["dpre"].items():
        worst["dz " + k] = rel(P.dz[k][..., :gref.shape[-1]], gref)
    for k, gref in o["grads"].items():
        if k == "aspp_1x1/weights":
            worst[k] = rel(eng.grads[k][:, :, :768], gref[:, :, :768])      # the aspp_4 rows multiply rounding noise
        else:
            worst[k] = rel(eng.grads[k], gref)
    print({k: "%.1e" % v for k, v in worst.items()})
    for k, e in worst.items():
        assert e < 1e-3, (k, e)
    assert abs(eng.losses(P)[0] - float(o["model_loss"])) < 1e-5 * abs(float(o["model_loss"]))
