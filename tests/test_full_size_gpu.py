"""BASELINE.json configs[1] at its FULL size - 32 synthetic documents, 64x64 grid, vocabulary 20000, embedding 128,
18 classes, the very batch bench.py times - checked through what does not need the oracle on all 32 documents:

  * grid build: the HIP builder against oracle/grid_oracle.py on all 32 documents, bit for bit (the grid oracle
    takes ~1 ms a document);
  * sample independence: the instance norm is per sample, so the logits of documents 0 and 1 inside the batch of
    32 must equal, bit for bit, those of the same two documents run as a batch of 2 (different launch geometry:
    the heaviest-first tile order applies to batches that are a multiple of 8 only) - and THOSE are compared with
    the float64 oracle (f32 path, 1e-4 of the largest logit);
  * per-(sample, channel) moments of a normalised layer: mean 0, variance 1 (or 0 for a dead channel);
  * softmax rows sum to 1; the loss the device reports equals model_cutie.py:116-130 restated on the host over the
    device's own per-cell cross-entropies (hard-negative mining over all 131072 cells);
  * one Adam step: a first step moves every parameter by at most the learning rate (main_train_json.py:152-156).

The model oracle is PARITY UNPINNED against TensorFlow (oracle/cutie_oracle.py header).
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cutie_oracle as co  # noqa: E402
from oracle import grid_oracle as go  # noqa: E402

DEV = "cuda:0"
V, E, C, H, W, B = 20000, 128, 18, 64, 64, 32


@pytest.fixture(scope="module")
def batch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cutie_amd import ops
    from cutie_amd.synthetic import as_token_dicts, make_documents
    packed = make_documents(B, C, seed=1234)
    tok = {k: torch.from_numpy(v).to(DEV) for k, v in packed.items()}
    g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], H, W, 1)
    torch.cuda.synchronize()
    return dict(packed=packed, docs=as_token_dicts(packed), g=g)


def _engine(dtype):
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIERes(V, C, p)
    net.build_loss()
    return net, Engine(net, DEV, dtype=dtype, seed=0)


def test_grid_build_of_the_bench_batch_is_bit_exact(batch):
    ref = go.positional_mapping(batch["docs"], H, W, 1, C)
    g = batch["g"]
    assert not g["status"].any()
    np.testing.assert_array_equal(g["grid"].cpu().numpy(), ref["grid"])
    np.testing.assert_array_equal(g["labels"].cpu().numpy(), ref["labels"])
    np.testing.assert_array_equal(g["ps_x"].cpu().numpy(), ref["ps_x"])
    np.testing.assert_array_equal(g["ps_y"].cpu().numpy(), ref["ps_y"])
    filled = int((g["grid"] != 0).sum())
    assert 32 * 150 < filled < 32 * 700        # U{200..600} tokens a document, a few dropped on crowded rows


def test_full_batch_f32_sample_independence_and_oracle(batch):
    net, eng = _engine("f32")
    grid, labels = batch["g"]["grid"], batch["g"]["labels"].to(torch.int32)
    labels = labels.clamp(0, C - 1)
    P = eng.forward(grid, labels=labels)
    eng.check_inputs(P)
    logits32 = eng.get(P, "cls_logits")[..., :C].clone()
    probs = P.probs.clone()
    ce = P.ce.clone()
    # softmax rows
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-5
    # moments of two normalised layers over all 32 samples
    for name in ("encoder1_8", "aspp_1x1"):
        y = eng.get(P, name).double()
        mean = y.mean(dim=(1, 2))
        var = y.var(dim=(1, 2), unbiased=False)
        assert float(mean.abs().max()) < 1e-4, name
        live = var > 1e-3
        assert float((var[live] - 1).abs().max()) < 2e-3, name
        assert (~live).sum() == 0 or float(var[~live].abs().max()) < 1e-3, name
    # the same two documents as a batch of 2
    P2 = eng.forward(grid[:2], labels=labels[:2])
    logits2 = eng.get(P2, "cls_logits")[..., :C]
    assert torch.equal(logits2, logits32[:2])
    # and against the float64 oracle
    params = {k: torch.from_numpy(np.asarray(v)).double() for k, v in eng.state_dict().items()}
    L = co.aspp_forward(params, grid[:2].cpu().long())
    ref = L["cls_logits"]
    err = float((logits2.double().cpu() - ref).abs().max() / ref.abs().max())
    print("full-size f32 logits vs oracle (docs 0-1): %.2e" % err)
    assert err < 1e-4
    agree = float((logits2.argmax(-1).cpu() == ref.argmax(-1)).double().mean())
    assert agree >= 0.999
    # the loss over all 131072 cells: model_cutie.py:116-130 on the device's own cross-entropies
    eng.forward(grid, labels=labels)
    eng.loss_backward(P)
    ml, rg, tl = eng.losses(P)
    loss_ref, w_ref, n_fg, n_bg = co.hnm_loss(ce.double().cpu(), labels.cpu(), 3)
    assert n_fg > 0 and n_bg == min(int((labels == 0).sum()), 3 * n_fg)
    assert abs(ml - float(loss_ref)) < 1e-5 * abs(float(loss_ref))
    assert int(((P.weight.cpu() != 0) != (w_ref != 0)).sum()) == 0
    reg_ref = sum(5e-4 * 0.5 * float((p ** 2).sum()) for n, p in params.items()
                  if n.endswith("/weights") and n != "embedding/weights")
    assert abs(rg - reg_ref) < 1e-4 * reg_ref
    # the gradient of the logits layer's bias is the sum of dz over all cells: both sides from device buffers
    dz = P.dz["cls_logits"][..., :C].double().sum(dim=(0, 1, 2)).cpu()
    gb = eng.grads["cls_logits/biases"].double().cpu()
    assert float((dz - gb).abs().max()) <= 1e-5 * float(dz.abs().max()) + 1e-9


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_full_batch_16bit_step(batch, dtype):
    """The timed configuration itself: sample independence bit for bit, reproducible step, bounded first Adam move."""
    from cutie_amd.train import Trainer
    net, eng = _engine(dtype)
    grid, labels = batch["g"]["grid"], batch["g"]["labels"].to(torch.int32).clamp(0, C - 1)
    P = eng.forward(grid, labels=labels)
    logits32 = eng.get(P, "cls_logits")[..., :C].clone()
    P8 = eng.forward(grid[8:16], labels=labels[8:16])
    assert torch.equal(eng.get(P8, "cls_logits")[..., :C], logits32[8:16])
    P3 = eng.forward(grid[5:8], labels=labels[5:8])          # not a multiple of 8: row-major tile order
    assert torch.equal(eng.get(P3, "cls_logits")[..., :C], logits32[5:8])
    before = eng.params_flat.clone()
    tr = Trainer(eng, learning_rate=1e-4)
    out = tr.step({"grid_table": grid, "gt_classes": labels})
    assert np.isfinite(out["total_loss"]) and out["total_loss"] > 0
    move = (eng.params_flat - before).abs()
    assert float(move.max()) <= 1.01e-4                       # |first Adam step| <= lr
    conv = slice(0, eng.n_conv_w)
    assert float((move[conv] > 0.5e-4).double().mean()) > 0.8  # and nearly all conv weights move by ~lr
    # the same step from the same start is the same step: conv kernels and biases bit for bit
    after = eng.params_flat.clone()
    eng.params_flat.copy_(before)
    eng.repack()
    tr2 = Trainer(eng, learning_rate=1e-4)
    out2 = tr2.step({"grid_table": grid, "gt_classes": labels})
    assert out2["total_loss"] == out["total_loss"]
    if out["global_norm"] < 9.9:      # below the clip norm the embedding gradient's atomics cannot reach the other updates
        assert torch.equal(eng.params_flat[:eng.n_dense], after[:eng.n_dense])


def test_training_trajectory_one_launch_vs_two_call_vs_f32(batch):
    """Twelve Adam steps on 8 documents of the bench batch from one start, three ways: f32 (conv, norm and their
    gradients as separate exact passes), f16 with every layer in its two-call form, f16 as shipped (forward and data
    gradient in one launch each, the backward pass reading y only).  The loss falls on all three and the 16-bit
    trajectories stay on the f32 one - an error in the ReLU pattern carried inside y, or in a tile meeting, would
    show as a drift within a few steps."""
    from cutie_amd.train import Trainer
    grid, labels = batch["g"]["grid"][:8], batch["g"]["labels"][:8].to(torch.int32).clamp(0, C - 1)
    runs = {}
    start = None
    for key, dtype, fused in (("f32", "f32", False), ("f16 two-call", "f16", False), ("f16", "f16", True)):
        net, eng = _engine(dtype)
        if start is None:
            start = eng.params_flat.clone()
        eng.params_flat.copy_(start)
        eng.repack()
        eng.train_fwd_fused = fused
        tr = Trainer(eng, learning_rate=1e-3, check_inputs=True)
        losses = [tr.step({"grid_table": grid, "gt_classes": labels})["model_loss"] for _ in range(12)]
        if dtype == "f16":
            P = eng.forward(grid, labels=labels)
            assert any(P.fused.values()) == fused
            assert eng.loss_scale_value()[1] == 0                     # no step was skipped for an overflow
        runs[key] = np.asarray(losses, dtype=np.float64)
        del eng, tr
        torch.cuda.empty_cache()
    for key, l in runs.items():
        print("%-13s model loss by step: %s" % (key, " ".join("%.4f" % v for v in l)))
        assert np.all(np.isfinite(l)) and l[-1] < 0.8 * l[0], key
    ref = runs["f32"]
    for key in ("f16 two-call", "f16"):
        rel = np.abs(runs[key] - ref) / ref
        assert rel[0] < 5e-3 and rel.max() < 6e-2, (key, rel)
    assert np.abs(runs["f16"] - runs["f16 two-call"]).max() / ref.max() < 6e-2
