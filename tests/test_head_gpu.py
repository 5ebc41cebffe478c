"""The logits head (cutie_head_fwd / cutie_head_bwd) against the f64 oracle, kernel level, through the C ABI.

Reference arithmetic: Model.conv(1, 1, C) -> bias -> ReLU -> instance norm (model_cutie_aspp.py:72-73,
model_framework.py:414-424), Model.softmax (:318-320), CUTIE.build_loss (model_cutie.py:107-130).  Both sides see the SAME
16-bit inputs (x and the weights rounded to the activation dtype), so only accumulation order and the final 16-bit
rounding of dz differ: the f32 outputs are held to 1e-5 .. 1e-4, dz to a 16-bit ulp."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cutie_oracle as co  # noqa: E402

DEV = "cuda:0"
TDT = {"bf16": torch.bfloat16, "f16": torch.float16}


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _case(B, H, W, cin, C, dtype, seed, fg_share=0.1):
    g = torch.Generator().manual_seed(seed)
    xw = torch.randn((B, H, W, cin + 64), generator=g).to(TDT[dtype])           # the layer's input is a channel slice
    w = (torch.randn((1, 1, cin, C), generator=g) * 0.15)
    bias = torch.randn(C, generator=g) * 0.2
    labels = torch.zeros((B, H, W), dtype=torch.int32)
    fg = torch.rand((B, H, W), generator=g) < fg_share
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g, dtype=torch.int32)
    return xw, w, bias, labels


def _run(ops, xw, w, bias, labels, cin, C, ratio=3, loss_scale=None, lab_dtype=torch.int32):
    B, H, W = labels.shape
    HW = H * W
    x = xw.to(DEV)[..., :cin]
    Cp = (C + 3) // 4 * 4
    out = dict(
        a=torch.full((B, H, W, Cp), 7.0, dtype=torch.float32, device=DEV),
        stats=torch.zeros((B, ops.head_tiles(HW) + 1, C, 2), dtype=torch.float32, device=DEV),
        logits=torch.full((B, H, W, Cp), 7.0, dtype=torch.float32, device=DEV),
        probs=torch.zeros((B * HW, C), dtype=torch.float32, device=DEV),
        ce=torch.zeros(B * HW, dtype=torch.float32, device=DEV),
        bad=torch.zeros(1, dtype=torch.int32, device=DEV),
        ws=torch.empty(ops.head_workspace_bytes(B, HW), dtype=torch.uint8, device=DEV),
        weight=torch.zeros(B * HW, dtype=torch.float32, device=DEV),
        red=torch.zeros((B, ops.head_chunks(HW), C, 2), dtype=torch.float32, device=DEV),
        loss=torch.zeros(4, dtype=torch.float32, device=DEV),
        dz=torch.full((B, H, W, 64), 3.0, dtype=xw.dtype, device=DEV),
        dbias=torch.zeros((B * ops.head_chunks(HW), C), dtype=torch.float32, device=DEV),
    )
    lab = labels.to(lab_dtype).to(DEV)
    wd, bd = w.to(DEV), bias.to(DEV)
    ops.head_fwd(x, cin, wd, bd, C, out["a"], out["stats"], out["logits"], probs=out["probs"], labels=lab, ce=out["ce"],
                 bad=out["bad"], workspace=out["ws"])
    ops.head_bwd(out["a"], out["stats"], out["logits"], out["probs"], lab, out["ce"], C, bd, ratio, out["weight"], out["red"],
                 out["loss"], out["dz"], out["dbias"], out["ws"], loss_scale=loss_scale)
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cutie_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("B,H,W,cin,C,labdt", [(3, 40, 37, 64, 18, torch.int32), (2, 64, 64, 64, 18, torch.int8),
                                                (2, 19, 23, 128, 9, torch.int32), (1, 16, 80, 64, 5, torch.int8),
                                                (2, 33, 31, 64, 32, torch.int32)])
def test_head_forward_and_backward_against_the_oracle(ops, dtype, B, H, W, cin, C, labdt):
    xw, w, bias, labels = _case(B, H, W, cin, C, dtype, seed=B * 1000 + H + C)
    S = 512.0
    ls = ops.loss_scale_state(S, DEV)
    o = _run(ops, xw, w, bias, labels, cin, C, loss_scale=ls, lab_dtype=labdt)
    # ---- oracle, f64, from the same rounded operands
    x64 = xw[..., :cin].double()
    w64 = w.to(TDT[dtype]).double().requires_grad_(False)
    z = (x64.reshape(-1, cin) @ w64.reshape(cin, C)).reshape(B, H, W, C) + bias.double()
    z.requires_grad_(True)
    mask = (o["a"][..., :C] > 0).double().cpu()
    a = z * mask
    y = co.instance_norm(a)
    probs = torch.softmax(y, dim=-1)
    ce = co.softmax_ce(y.reshape(-1, C), labels.reshape(-1))
    Cp = (C + 3) // 4 * 4
    assert relerr(o["a"][..., :C], torch.relu(z)) < 1e-5
    flips = int(((z.detach() > 0) != (mask > 0)).sum())
    assert flips <= 2, "%d ReLU decisions differ from the f64 pre-activation" % flips
    assert relerr(o["logits"][..., :C], y) < 1e-4
    if Cp > C:
        assert float(o["a"][..., C:].abs().max()) == 0.0 and float(o["logits"][..., C:].abs().max()) == 0.0      # padding classes
    assert relerr(o["probs"], probs.reshape(-1, C)) < 1e-4
    assert float((o["ce"].double().cpu() - ce.detach()).abs().max()) < 1e-4
    assert int(o["bad"]) == 0
    # moments: slot sums about relu(bias) (K6), the slot beyond head_tiles untouched
    sh = torch.relu(bias.double())
    st = o["stats"].double().cpu()
    assert float(st[:, -1].abs().max()) == 0.0
    d = torch.relu(z.detach()) - sh
    assert relerr(st.sum(dim=1)[..., 0], d.sum(dim=(1, 2))) < 1e-5 and relerr(st.sum(dim=1)[..., 1], (d * d).sum(dim=(1, 2))) < 1e-5
    # ---- hard negative mining on the DEVICE's CE (a top-k boundary between two CE values an ulp apart is no kernel error)
    loss_ref, w_ref, n_fg, num_bg = co.hnm_loss(o["ce"].double().cpu(), labels, 3)
    assert torch.equal(o["weight"].cpu() != 0, w_ref != 0)
    assert relerr(o["weight"], w_ref) < 1e-6
    lo = o["loss"].cpu().tolist()
    assert abs(lo[0] - float(loss_ref)) < 1e-5 * abs(float(loss_ref)) and lo[1] == n_fg and lo[2] == num_bg
    # ---- backward: d(sum w * ce) / d(pre-activation) with the device's selection and ReLU pattern
    (w_ref * ce).sum().backward()
    dz_ref = z.grad
    dz = o["dz"][..., :C].double().cpu() / S
    tol = 2.0 ** (-10 if dtype == "f16" else -7)
    assert relerr(dz, dz_ref) < tol, relerr(dz, dz_ref)
    assert float(o["dz"][..., Cp:].float().abs().min()) == 3.0 and float(o["dz"][..., Cp:].float().abs().max()) == 3.0      # left alone
    if Cp > C:
        assert float(o["dz"][..., C:Cp].float().abs().max()) == 0.0
    db = o["dbias"].double().cpu().sum(dim=0) / S
    assert relerr(db, o["dz"][..., :C].double().cpu().sum(dim=(0, 1, 2)) / S) < 5 * tol   # sums of the f32 values the stored dz was rounded from
    assert relerr(db, dz_ref.sum(dim=(0, 1, 2))) < 5 * tol


def test_head_edge_cases(ops):
    """no foreground cell -> NaN loss (tf.reduce_mean of an empty tensor) and zero weights' gradient; fewer background cells
    than ratio * foreground -> all of them selected; a label outside [0, C) is counted and gives a NaN CE; exact CE ties are
    taken lower index first (tf.nn.top_k)"""
    B, H, W, cin, C = 2, 32, 32, 64, 18
    xw, w, bias, labels = _case(B, H, W, cin, C, "f16", seed=77)
    o = _run(ops, xw, w, bias, torch.zeros_like(labels), cin, C)
    assert np.isnan(o["loss"].cpu().tolist()[0]) and float(o["weight"].abs().max()) == 0.0
    many = labels.clone()
    many[:, :, :20] = 1                                      # 62 % foreground: 3 * Nfg > Nbg
    o = _run(ops, xw, w, bias, many, cin, C)
    wv = o["weight"].cpu().reshape(B, H, W)
    n_fg, n_bg = int((many != 0).sum()), int((many == 0).sum())
    assert torch.all(wv[many == 0] == 1.0 / n_bg) and abs(float(wv[many != 0][0]) - 1.0 / n_fg) < 1e-9
    badl = labels.clone()
    badl[0, 0, 0], badl[1, 5, 5] = C, -3
    o = _run(ops, xw, w, bias, badl, cin, C)
    assert int(o["bad"]) == 2 and np.isnan(float(o["ce"][0]))
    # ties: identical cells (identical inputs) give identical CE; only the first k of them may be selected
    xt = xw.clone()
    xt[:] = xw[0, 0, 0].clone()
    xt[0, :2] = xw[0, :2]                                    # a few distinct rows so that the norm is not degenerate
    lt = torch.zeros_like(labels)
    lt[0, 0, :8] = 2                                         # 8 foreground -> 24 background cells selected
    o = _run(ops, xt, w, bias, lt, cin, C)
    ce = o["ce"].double().cpu()
    _loss, w_ref, _nfg, num_bg = co.hnm_loss(ce, lt, 3)
    assert num_bg == 24 and torch.equal(o["weight"].cpu() != 0, w_ref != 0)


def test_head_inference_mode_matches_training_forward(ops):
    B, H, W, cin, C = 2, 30, 50, 64, 18
    xw, w, bias, labels = _case(B, H, W, cin, C, "f16", seed=5)
    o = _run(ops, xw, w, bias, labels, cin, C)
    x = xw.to(DEV)[..., :cin]
    Cp = 20
    a = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=DEV)
    st = torch.zeros((B, ops.head_tiles(H * W) + 1, C, 2), dtype=torch.float32, device=DEV)
    lg = torch.zeros_like(a)
    pr = torch.zeros((B * H * W, C), dtype=torch.float32, device=DEV)
    ops.head_fwd(x, cin, w.to(DEV), bias.to(DEV), C, a, st, lg, probs=pr)
    torch.cuda.synchronize()
    assert torch.equal(lg, o["logits"]) and torch.equal(pr, o["probs"])
