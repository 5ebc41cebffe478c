"""End-to-end parity of the HIP engine against oracle/cutie_oracle.py (float64 on PyTorch CPU):
every named layer of CUTIE-ASPP, logits, softmax, argmax agreement, the three losses, every
gradient tf.gradients would produce, the TF-style global norm and one clipped Adam step.

Tolerances (BASELINE.json north_star): f32 path - every layer and the logits within 1e-4,
measured as max |got - ref| / max |ref| (the strictest reading).  f16 path - the HEADLINE 16-bit path
(Engine's and bench.py's default): every layer and the logits within north_star's 16-bit bound 2e-2 in
both rel-L2 and max-norm, every gradient within 3e-2 rel-L2.  bf16 is a non-default option of the same
kernels: its 8-bit significand puts ~5e-3 relative L2 error on every layer's normalised output (instance
norm re-scales each channel to unit variance, so nothing is attenuated) and fourteen layers compound it to
~3.8e-2 on the logits - OUTSIDE the tolerance, which is why it is not the headline; its bound here is the
measured level with margin so that regressions show.  Per-cell argmax agreement >= 99.9 % on the cells
whose oracle top-2 margin exceeds the path's logit tolerance (the share of such cells and the
agreement over ALL cells are printed; the f32 path must reach 99.9 % over all cells).
The oracle is PARITY UNPINNED against TensorFlow (see its header); the aspp_4 slice is
compared with an absolute tolerance (it is rounding noise around zero).
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cutie_oracle as co  # noqa: E402

DEV = "cuda:0"
V, E, C = 600, 128, 18


def make_inputs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    grid = torch.zeros((B, H, W), dtype=torch.int64)
    occ = torch.rand((B, H, W), generator=g) < 0.12
    ids = torch.randint(2, V, (B, H, W), generator=g)
    ids[torch.rand((B, H, W), generator=g) < 0.35] = 1
    grid[occ] = ids[occ]
    labels = torch.zeros((B, H, W), dtype=torch.int64)
    fg = occ & (torch.rand((B, H, W), generator=g) < 0.25)
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g)
    return grid, labels


def params_ns():
    return types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=2)


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def oracle_run():
    torch.manual_seed(0)
    B, H, W = 2, 64, 64
    grid, labels = make_inputs(B, H, W, 3)
    params = co.init_params(V, E, C, seed=0)
    out = co.loss_and_grads(params, grid, labels)
    return dict(B=B, H=H, W=W, grid=grid, labels=labels, params=params, out=out)


HEADLINE_TOL = 2e-2          # north_star's 16-bit logit tolerance; the f16 path must meet it everywhere
BF16_OPTION_TOL = 6e-2       # the non-default bf16 option: measured 3.8e-2 (14 layers) with margin, NOT a parity claim
BF16_OPTION_TOL_DEEP = 1.5e-1   # the same over CUTIE2's 21 / 26 layers (measured 1.1e-1)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_cutie_aspp_forward_backward_parity(oracle_run, dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    r = oracle_run
    tol = {"f32": 1e-4, "f16": HEADLINE_TOL, "bf16": BF16_OPTION_TOL}[dtype]
    net = CUTIERes(V, C, params_ns())
    fetches = net.build_loss()
    assert [f.name for f in fetches] == ["model_loss", "regularization_loss", "total_loss", "cls_logits", "softmax"]
    eng = Engine(net, DEV, dtype=dtype)
    eng.load_params(r["params"])
    P = eng.forward(r["grid"].unsqueeze(-1), labels=r["labels"])
    eng.check_inputs(P)
    L = r["out"]["layers"]
    report = {}
    for name in ("embedding", "encoder1_1", "encoder1_4", "encoder1_8", "aspp_1", "aspp_3", "aspp_1x1",
                 "decoder1_1", "cls_logits", "softmax"):
        report[name] = rel(eng.get(P, name), L[name])
    print(dtype, "layer errors", {k: "%.2e" % v for k, v in report.items()})
    assert report["embedding"] <= {"f32": 1e-7, "bf16": 4e-3, "f16": 5e-4}[dtype]   # rounding of the table
    report2 = {name: rel2(eng.get(P, name), L[name])[0] for name in report}
    print(dtype, "layer rel-L2 errors", {k: "%.2e" % v for k, v in report2.items()})
    for name, e in report.items():
        if dtype == "f32":
            assert e < tol, (name, e)
        elif dtype == "f16":      # the north-star 16-bit bound, in both norms
            assert report2[name] < HEADLINE_TOL and e < HEADLINE_TOL, (name, report2[name], e)
        else:
            assert report2[name] < BF16_OPTION_TOL and e < 8e-2, (name, report2[name], e)
    # aspp_4 is the spatial mean of an instance-normalised tensor: ~0 on both sides
    a4 = eng.get(P, "aspp_4").float().abs().max().item()
    assert a4 < (1e-5 if dtype == "f32" else 2e-3) and float(L["aspp_4"].abs().max()) < 1e-9
    # argmax agreement
    ref_logits = L["cls_logits"].reshape(-1, C)
    got_arg = eng.get(P, "cls_logits").reshape(-1, C).argmax(-1).cpu()
    ref_arg = ref_logits.argmax(-1)
    top2 = ref_logits.topk(2, dim=-1).values
    margin = top2[:, 0] - top2[:, 1]
    agree = (got_arg == ref_arg).double().mean().item()
    decided = margin > tol * float(ref_logits.abs().max())
    agree_decided = (got_arg == ref_arg)[decided].double().mean().item()
    print(dtype, "argmax agreement %.5f (all cells) %.5f (cells with margin > tol: %.4f of cells)"
          % (agree, agree_decided, decided.double().mean().item()))
    assert agree_decided >= 0.999
    if dtype == "f32":
        assert agree >= 0.999
    S = eng.loss_scale_value()[0]
    assert (S > 1.0) == (dtype == "f16")
    # loss + backward.  Gradients are compared on the device's own ReLU activation pattern (see
    # oracle cbrn(mask=...)); how many cells sit on the other side of a kink in f64 is reported.
    eng.loss_backward(P)
    ml, rg, tl = eng.losses(P)
    conv_names = [n.name for n in net.nodes if n.op == "conv"]
    masks = {k: eng.relu_mask(P, k).double().cpu() for k in conv_names}
    flips = sum(int(((L[k] - L[k].amin(dim=(1, 2), keepdim=True)).abs() < 1e-12).ne(masks[k] == 0).sum())
                for k in conv_names)
    total = sum(masks[k].numel() for k in conv_names)
    print(dtype, "ReLU pattern: %d of %d activations differ from the f64 oracle's" % (flips, total))
    # a pre-activation within rounding of zero may fall on either side; anything beyond that share would be a wrong
    # pattern (e.g. one carried wrongly inside y): f32 measured 46 of 25.8 M = 1.8e-6, 16-bit paths ~1e-3 (their
    # pre-activations carry ~5e-3 relative error, and ~0.1 % of them lie that close to zero)
    assert flips <= {"f32": 1e-5, "f16": 3e-3, "bf16": 2e-2}[dtype] * total, (flips, total)
    o = co.loss_and_grads(r["params"], r["grid"], r["labels"], masks=masks, cell_weights=P.weight.double().cpu())
    assert abs(ml - float(r["out"]["model_loss"])) < tol * abs(float(r["out"]["model_loss"]))
    assert abs(rg - float(o["reg_loss"])) < 1e-4 * abs(float(o["reg_loss"]))
    _, w_ref, _, _ = co.hnm_loss(co.softmax_ce(o["layers"]["cls_logits"].reshape(-1, C), r["labels"].reshape(-1)).detach(),
                                 r["labels"], 3)
    sel_diff = int(((P.weight.cpu() != 0) != (w_ref != 0)).sum())
    print(dtype, "hard-negative selection: cells that differ from the oracle's:", sel_diff)
    # gradient of every conv's pre-activation (the device's dz; the one-launch data gradient writes no dy) and of
    # the embedding output
    derr = {k: rel(P.dz[k][..., :g.shape[-1]].double() / S, g) for k, g in o["dpre"].items()}
    derr["embedding"] = rel(P.dy["embedding"][..., :E].double() / S, o["dlayers"]["embedding"])
    print(dtype, "d(layer) errors", {k: "%.2e" % v for k, v in derr.items()})
    gerr = {k: rel(eng.grads[k], g) for k, g in o["grads"].items()}
    gerr["aspp_1x1/weights"] = rel(eng.grads["aspp_1x1/weights"][:, :, :768], o["grads"]["aspp_1x1/weights"][:, :, :768])
    print(dtype, "grad errors", {k: "%.2e" % v for k, v in gerr.items()})
    g2 = {k: rel2(eng.grads[k], g)[0] for k, g in o["grads"].items() if k != "aspp_1x1/weights"}
    print(dtype, "grad rel-L2 errors", {k: "%.2e" % v for k, v in g2.items()})
    if dtype == "f32":
        assert sel_diff == 0
        for k, e in list(derr.items()) + list(gerr.items()):
            assert e < 1e-3, (k, e)
    elif dtype == "f16":   # gradients come out of the loss scale again; measured ~5e-3
        for k, e in g2.items():
            assert e < 3e-2, (k, e)
    else:   # bf16: measured ~2e-2..6e-2 rel-L2 (activations carry ~3e-2 by the last layers)
        for k, e in g2.items():
            assert e < 1.5e-1, (k, e)
    gn = math_sqrt(eng.global_norm_sq(P, tf_indexed_slices=True))
    gn_ref = float(co.global_norm_tf(o["grads"], o["d_embedding_out"]))
    assert abs(gn - gn_ref) < {"f32": 1e-3, "bf16": 1e-1, "f16": 2e-2}[dtype] * gn_ref
    # one Adam step through the Trainer == oracle step
    eng.load_params(r["params"])
    tr = Trainer(eng, learning_rate=1e-4)
    vals = tr.step({"grid_table": r["grid"].unsqueeze(-1), "gt_classes": r["labels"]})
    assert abs(vals["total_loss"] - float(r["out"]["total_loss"])) < tol * abs(float(r["out"]["total_loss"]))
    state = {}
    newp = co.adam_step_tf(r["params"], o["grads"], state, 1e-4, gn_ref)
    # Adam's first step moves every weight by ~lr*sign(g): compare the moves
    for k in ("encoder1_1/weights", "decoder1_1/weights", "cls_logits/biases"):
        move_ref = (newp[k] - r["params"][k])
        move = eng.params[k].double().cpu() - r["params"][k]
        big = o["grads"][k].abs() > 0.02 * o["grads"][k].abs().max()     # away from g ~ 0, where sign(g) is noise
        frac_same_sign = float((torch.sign(move) == torch.sign(move_ref))[big].double().mean())
        assert frac_same_sign > {"f32": 0.9999, "bf16": 0.95, "f16": 0.99}[dtype], (k, frac_same_sign)
        assert float(move[big].abs().max()) < 1.01e-4      # |first Adam step| <= lr


def rel2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = (got - ref)
    return (float(d.norm() / ref.norm().clamp_min(1e-30)),
            float((d.abs() > 1e-4 * ref.abs().max()).double().mean()))


def math_sqrt(t):
    return float(torch.sqrt(t.double()).item())


def test_shapes_other_than_64(oracle_run):
    """Odd grid shapes the loader produces under augmentation (rows/cols up to 80)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    net = CUTIERes(V, C, params_ns())
    net.build_loss()
    eng = Engine(net, DEV, dtype="f32")
    eng.load_params(oracle_run["params"])
    for (B, H, W) in [(1, 42, 63), (1, 70, 80)]:
        grid, labels = make_inputs(B, H, W, 5)
        L = co.aspp_forward(oracle_run["params"], grid)
        P = eng.forward(grid.unsqueeze(-1))
        assert rel(eng.get(P, "cls_logits"), L["cls_logits"]) < 1e-4
        assert rel(eng.get(P, "softmax"), L["softmax"]) < 1e-4


def test_plan_cache_is_bounded_over_many_grid_shapes(oracle_run):
    """The reference's loader draws a new (rows, cols) for almost every training batch (data_loader_json.py:660-694);
    the engine keeps buffer sets for the `max_plans` most recently used shapes only."""
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    net = CUTIERes(V, C, params_ns())
    net.build_loss()
    eng = Engine(net, DEV, dtype="f16", max_plans=3)
    tr = Trainer(eng)
    g = torch.Generator().manual_seed(1)
    peak = []
    shapes = [(int(torch.randint(42, 81, (1,), generator=g)), int(torch.randint(63, 81, (1,), generator=g))) for _ in range(20)]
    for k, (H, W) in enumerate(shapes):
        grid, labels = make_inputs(2, H, W, 100 + k)
        out = tr.step({"grid_table": grid.unsqueeze(-1), "gt_classes": labels})
        assert np.isfinite(out["total_loss"])
        torch.cuda.synchronize()
        peak.append(torch.cuda.memory_allocated())
        assert len(eng._plans) <= 3
    # after the cache has filled, memory stays level: the 20th shape needs no more than the worst three of them
    assert max(peak[3:]) <= 1.6 * max(peak[:3]), (peak[:3], max(peak))
    # a shape seen again while still cached reuses its plan
    P = eng._plans[next(reversed(eng._plans))]
    H, W = shapes[-1]
    grid, labels = make_inputs(2, H, W, 7)
    assert eng.forward(grid.unsqueeze(-1), labels=labels) is P


# ------------------------------------------------------------------------------ CUTIE2 (configs 4 / 5): forward
@pytest.mark.parametrize("variant,B,H,W,Hi", [("aspp", 2, 32, 48, 40), ("dilate", 1, 48, 32, 56)])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_cutie2_forward_parity(variant, B, H, W, Hi, dtype):
    """CUTIE2-ASPP / CUTIE2-dilate inference: image branch (3x3 convs, its own ASPP), positional
    sampling of the image feature map at the token pixels, 1x1 fuser, the CUTIE-ASPP trunk.
    Oracle: oracle/cutie2_oracle.py (float64, PARITY UNPINNED against TensorFlow)."""
    from oracle import cutie2_oracle as c2
    from cutie_amd.runtime import Engine
    if variant == "aspp":
        from cutie_amd.model_cutie2_aspp import CUTIE2
    else:
        from cutie_amd.model_cutie2_dilate import CUTIE2
    graph = c2.GRAPHS[variant]
    g = torch.Generator().manual_seed(11)
    grid, _ = make_inputs(B, H, W, 5)
    image = torch.rand((B, Hi, W, 3), generator=g) - 0.5
    ps = torch.randint(0, Hi * W - 1, (B, H * W), generator=g)
    params = c2.init_params(graph, V, E, 9, seed=2)
    ref = c2.forward(graph, params, grid, image.double(), ps)

    net = CUTIE2(V, 9, params_ns())
    assert [s.name for s in net.param_specs] == list(params.keys())           # creation order
    assert [tuple(s.shape) for s in net.param_specs] == [tuple(v.shape) for v in params.values()]
    eng = Engine(net, DEV, dtype=dtype)
    eng.load_params(params)
    P = eng.forward(grid.unsqueeze(-1), image=image, ps_1d_indices=ps)
    torch.cuda.synchronize()
    report = {}
    for op, name, _inputs, _args in graph:
        if op in ("conv", "embed", "positional_sampling", "softmax"):
            got = eng.get(P, name).float().cpu()
            r = ref[name]
            report[name] = float((got.double() - r).pow(2).sum().sqrt() / r.pow(2).sum().sqrt().clamp_min(1e-30))
    print(variant, dtype, "rel-L2 per layer", {k: "%.1e" % v for k, v in report.items()})
    # bf16: the same ~5e-3 per layer as CUTIE-ASPP, compounded over 21 / 26 layers (see the module
    # docstring); the bound is the measured level with margin, not the 2e-2 target
    tol = {"f32": 1e-4, "f16": HEADLINE_TOL, "bf16": BF16_OPTION_TOL_DEEP}[dtype]
    for name, e in report.items():
        assert e < tol, (name, e)
    if dtype in ("f32", "f16"):
        agree = float((eng.get(P, "softmax").cpu().argmax(-1) == ref["softmax"].argmax(-1)).float().mean())
        print(variant, dtype, "argmax agreement over all cells %.5f" % agree)
    if dtype == "f32":
        assert rel(eng.get(P, "cls_logits")[..., :9], ref["cls_logits"]) < 1e-4
        agree = float((eng.get(P, "softmax").cpu().argmax(-1) == ref["softmax"].argmax(-1)).float().mean())
        assert agree >= 0.999
    # the pooled branches are rounding noise around zero (mean of an instance-normalised tensor)
    assert float(eng.get(P, "aspp_4").abs().max()) < (1e-5 if dtype == "f32" else 2e-2)


def test_cutie2_needs_its_image_inputs():
    from cutie_amd.model_cutie2_dilate import CUTIE2
    from cutie_amd.runtime import Engine
    from cutie_amd.capi import CutieError
    eng = Engine(CUTIE2(V, 9, params_ns()), DEV, dtype="f32")
    grid, _ = make_inputs(1, 16, 16, 1)
    with pytest.raises((CutieError, KeyError, TypeError, ValueError)):
        eng.forward(grid)


# ------------------------------------------------------------------------------ CUTIE2: loss + backward
@pytest.mark.parametrize("variant,B,H,W,Hi", [("aspp", 2, 32, 48, 40), ("dilate", 1, 48, 32, 56)])
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_cutie2_backward_parity(variant, B, H, W, Hi, dtype):
    """tf.gradients of CUTIE2's total loss (model_cutie2.py:114-147, main_train_json.py:152) through the
    fuser, the sampler (scatter-add into the image feature map, with most cells pointing at pixel 0
    as token-less cells do) and the 3x3 image branch.  Oracle: autograd over oracle/cutie2_oracle.py on
    the device's ReLU pattern and hard-negative selection (PARITY UNPINNED against TensorFlow)."""
    from oracle import cutie2_oracle as c2
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    if variant == "aspp":
        from cutie_amd.model_cutie2_aspp import CUTIE2
    else:
        from cutie_amd.model_cutie2_dilate import CUTIE2
    graph = c2.GRAPHS[variant]
    g = torch.Generator().manual_seed(12)
    grid, _ = make_inputs(B, H, W, 6)
    labels = torch.zeros((B, H, W), dtype=torch.int64)
    fg = (grid > 0) & (torch.rand((B, H, W), generator=g) < 0.3)
    labels[fg] = torch.randint(1, 9, (int(fg.sum()),), generator=g)
    image = torch.rand((B, Hi, W, 3), generator=g) - 0.5
    ps = torch.randint(1, Hi * W - 1, (B, H * W), generator=g)
    ps[(grid == 0).reshape(B, -1)] = 0                       # data_loader_json.py:554-555: no token -> pixel 0
    ps[:, 5] = ps[:, 6]                                      # two cells sampling one pixel
    params = c2.init_params(graph, V, E, 9, seed=2)

    net = CUTIE2(V, 9, params_ns())
    net.build_loss()
    eng = Engine(net, DEV, dtype=dtype)
    eng.load_params(params)
    P = eng.forward(grid.unsqueeze(-1), image=image, ps_1d_indices=ps, labels=labels)
    eng.loss_backward(P)
    torch.cuda.synchronize()
    S = eng.loss_scale_value()[0]
    conv_names = [n.name for n in net.nodes if n.op == "conv"]
    masks = {k: eng.relu_mask(P, k).double().cpu() for k in conv_names}
    o = c2.loss_and_grads(graph, params, grid, image.double(), ps, labels, masks=masks,
                          cell_weights=P.weight.double().cpu())
    ml, rg, tl = eng.losses(P)
    tol = {"f32": 1e-3, "f16": 3e-2}[dtype]
    assert abs(ml - float(o["model_loss"])) < tol * abs(float(o["model_loss"]))
    assert abs(rg - float(o["reg_loss"])) < 1e-4 * abs(float(o["reg_loss"]))
    sampled = "image_featuremap" if variant == "aspp" else "image_encoder1_8"
    dl = {k: rel2(P.dy[k][..., :o["dlayers"][k].shape[-1]].double() / S, o["dlayers"][k])[0]
          for k in ("positional_sampling", sampled, "embedding")}
    dl.update({"dz " + k: rel2(P.dz[k][..., :o["dpre"][k].shape[-1]].double() / S, o["dpre"][k])[0]
               for k in (sampled, "image_encoder1_1", "feature_fuser", "encoder1_4")})
    print(variant, dtype, "d(layer) rel-L2", {k: "%.1e" % v for k, v in dl.items()})
    gerr = {k: rel2(eng.grads[k], gr)[0] for k, gr in o["grads"].items()}
    print(variant, dtype, "grad rel-L2", {k: "%.1e" % v for k, v in gerr.items()})
    for k, e in list(dl.items()) + list(gerr.items()):
        assert e < tol, (k, e)
    # and one optimiser step runs end to end on the CUTIE2 batch dict
    tr = Trainer(eng, learning_rate=1e-4)
    out = tr.step({"grid_table": grid.unsqueeze(-1), "gt_classes": labels, "data_image": image, "ps_1d_indices": ps})
    assert np.isfinite(out["total_loss"]) and np.isfinite(out["global_norm"]) and out["global_norm"] > 0


# ------------------------------------------------------------------------------ use_ghm = 1
def test_cutie_aspp_ghm_loss_and_gradients(oracle_run):
    """sigmoid logits layer + gradient-harmonised loss (model_cutie_aspp.py:17,73; model_cutie.py:110;
    main_train_json.py:86-119,249-254) against the oracle, f32.  The weights are a fed constant in the
    reference's graph: the gradient check feeds the oracle the device's weights (and ReLU pattern and
    hard-negative selection), the weights themselves are checked against calc_ghm_weights restated."""
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    r = oracle_run
    pn = params_ns()
    pn.use_ghm, pn.ghm_bins, pn.ghm_momentum, pn.eps = 1, 30, 0, 1e-6
    net = CUTIERes(V, C, pn)
    net.build_loss()
    assert net.activation == "sigmoid" and net.layers["cls_logits"].attrs["activation"] == "sigmoid"
    eng = Engine(net, DEV, dtype="f32")
    eng.load_params(r["params"])
    P = eng.forward(r["grid"].unsqueeze(-1), labels=r["labels"])
    L = co.aspp_forward(r["params"], r["grid"], activation="sigmoid")
    # sigmoid of a ~1e-2 pre-activation is 0.5 + ~2e-3: f32 holds that signal to ~3e-5 of its spread before the
    # instance norm rescales it to unit variance - an f32 limit of the reference graph itself (measured 1.4e-4)
    assert rel(eng.get(P, "decoder1_1"), L["decoder1_1"]) < 1e-4
    assert rel(eng.get(P, "cls_logits"), L["cls_logits"]) < 5e-4
    assert rel(eng.get(P, "softmax"), L["softmax"]) < 5e-4             # of the UNweighted logits (model_cutie.py:138)
    eng.loss_backward(P)
    torch.cuda.synchronize()
    w_ref, _ = co.calc_ghm_weights(L["cls_logits"].float(), r["labels"], 30, 0, 1e-6)
    w_dev = P.ghm_w.reshape(w_ref.shape).cpu().double()
    # one element changing bins moves that bin's count, hence the weight of everything in it, by ~1e-3:
    # "another bin" is a weight that is off by more than the count noise
    differ = float(((w_dev - w_ref).abs() > 2e-2 * w_ref.abs()).double().mean())
    print("GHM weights: share of elements in another bin than the f64 oracle's logits put them: %.2e" % differ)
    assert differ < 5e-3 and abs(float(w_dev.mean() / w_ref.mean()) - 1) < 1e-2
    conv_names = [n.name for n in net.nodes if n.op == "conv"]
    masks = {k: eng.relu_mask(P, k).double().cpu() for k in conv_names}
    o = co.loss_and_grads(r["params"], r["grid"], r["labels"], masks=masks, cell_weights=P.weight.double().cpu(),
                          activation="sigmoid", ghm_weights=w_dev)
    ml, rg, tl = eng.losses(P)
    assert abs(ml - float(o["model_loss"])) < 1e-4 * abs(float(o["model_loss"]))
    gerr = {k: rel2(eng.grads[k], g)[0] for k, g in o["grads"].items()}
    print("GHM grad rel-L2", {k: "%.1e" % v for k, v in gerr.items()})
    for k, e in gerr.items():
        # the logits bias sits in front of sigmoid + instance norm: its gradient is a small difference of
        # large terms (the norm cancels a constant shift to first order), measured 1.7e-3; the rest ~1e-4
        assert e < (1e-2 if k == "cls_logits/biases" else 1e-3), (k, e)
    out = Trainer(eng, learning_rate=1e-4).step({"grid_table": r["grid"].unsqueeze(-1), "gt_classes": r["labels"]})
    assert np.isfinite(out["total_loss"]) and out["global_norm"] > 0
