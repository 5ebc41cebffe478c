"""Self-tests of oracle/cutie_oracle.py on the CPU (no GPU, no HIP library).

The model oracle is PARITY UNPINNED against TensorFlow 1.12 (nothing of the reference can run it here), so
it is pinned the only way left: every op of it against an INDEPENDENT statement of the same definition -
PyTorch's own instance-norm / cross-entropy / Adam kernels, numpy loops written from the TF op documentation,
and known-answer cases: single-tap impulse responses for every dilation, an all-zero channel
through the norm, Nfg = 0, the literal [Nfg,1] + [1,num_bg] broadcast of model_cutie.py:116-130.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cutie_oracle as co


def test_instance_norm_matches_torch_and_definition():
    g = torch.Generator().manual_seed(0)
    a = torch.randn((3, 9, 7, 5), generator=g, dtype=torch.float64) * 3 + 2
    a[..., 2] = 0.0                                   # an all-zero (dead ReLU) channel
    a[..., 3] = 7.5 + 1e-4 * torch.randn((3, 9, 7), generator=g, dtype=torch.float64)   # mean >> std
    got = co.instance_norm(a)
    ref = F.instance_norm(a.permute(0, 3, 1, 2), eps=1e-6).permute(0, 2, 3, 1)     # biased variance, no affine
    assert float((got - ref).abs().max()) < 1e-9
    assert float(got[..., 2].abs().max()) == 0.0      # (0 - 0) * rsqrt(0 + eps) = 0, not NaN
    # definition, one (sample, channel) by hand: model_framework.py:423, TF 1.12 epsilon 1e-6
    x = a[1, :, :, 3].numpy().reshape(-1)
    by_hand = (x - x.mean()) / np.sqrt(((x - x.mean()) ** 2).mean() + 1e-6)
    np.testing.assert_allclose(got[1, :, :, 3].numpy().reshape(-1), by_hand, rtol=0, atol=1e-10)
    # the spatial mean of a normalised tensor is zero (aspp_4 is rounding noise)
    assert float(co.global_pool(got).abs().max()) < 1e-12


@pytest.mark.parametrize("rate", [1, 2, 4, 8, 16])
def test_conv_single_tap_impulse_responses(rate):
    """tf.nn.conv2d / atrous_conv2d SAME, stride 1, HWIO: out[h, w, o] = sum_{i,j,c} x[h + (i-1)r, w + (j-2)r, c] W[i,j,c,o].
    A filter with ONE non-zero tap (i, j, c, o) must copy channel c, shifted by ((i-1)r, (j-2)r), into channel o,
    with zeros where the shifted pixel falls outside the image."""
    H, W, Cin, Cout = 40, 70, 3, 4
    g = torch.Generator().manual_seed(rate)
    x = torch.randn((1, H, W, Cin), generator=g, dtype=torch.float64)
    for (i, j, c, o) in [(0, 0, 0, 1), (1, 2, 1, 0), (2, 4, 2, 3), (0, 3, 1, 2), (2, 1, 0, 0)]:
        w = torch.zeros((3, 5, Cin, Cout), dtype=torch.float64)
        w[i, j, c, o] = 2.0
        y = co.conv_raw(x, w, rate)
        dy, dx = (i - 1) * rate, (j - 2) * rate
        want = torch.zeros((H, W), dtype=torch.float64)
        h0, h1 = max(0, -dy), min(H, H - dy)
        w0, w1 = max(0, -dx), min(W, W - dx)
        if h0 < h1 and w0 < w1:
            want[h0:h1, w0:w1] = 2.0 * x[0, h0 + dy:h1 + dy, w0 + dx:w1 + dx, c]
        assert torch.equal(y[0, :, :, o], want), (rate, i, j)
        others = [k for k in range(Cout) if k != o]
        assert float(y[0][..., others].abs().max()) == 0.0


def test_conv_against_numpy_loops():
    """the whole contraction, by loops, on a small case with a 1x1 and a 3x5 dilated filter"""
    g = torch.Generator().manual_seed(5)
    B, H, W, Cin, Cout, r = 2, 6, 9, 3, 2, 2
    x = torch.randn((B, H, W, Cin), generator=g, dtype=torch.float64)
    for kh, kw in ((3, 5), (1, 1), (3, 3)):
        w = torch.randn((kh, kw, Cin, Cout), generator=g, dtype=torch.float64)
        ref = np.zeros((B, H, W, Cout))
        xn, wn = x.numpy(), w.numpy()
        for b in range(B):
            for h in range(H):
                for ww in range(W):
                    for i in range(kh):
                        for j in range(kw):
                            hh, wj = h + (i - kh // 2) * r, ww + (j - kw // 2) * r
                            if 0 <= hh < H and 0 <= wj < W:
                                ref[b, h, ww] += xn[b, hh, wj] @ wn[i, j]
        np.testing.assert_allclose(co.conv_raw(x, w, r).numpy(), ref, rtol=0, atol=1e-12)


def test_cbrn_order_is_conv_bias_activation_norm():
    """model_framework.py:414-424: IN(act(conv + b)); the bias sits INSIDE the activation, the norm comes last"""
    g = torch.Generator().manual_seed(6)
    x = torch.randn((1, 8, 8, 4), generator=g, dtype=torch.float64)
    w = torch.randn((3, 5, 4, 6), generator=g, dtype=torch.float64) * 0.2
    b = torch.randn(6, generator=g, dtype=torch.float64)
    z = co.conv_raw(x, w, 1) + b
    assert torch.allclose(co.cbrn(x, w, b, 1), co.instance_norm(torch.relu(z)), atol=1e-12)
    assert torch.allclose(co.cbrn(x, w, b, 1, activation="sigmoid"), co.instance_norm(torch.sigmoid(z)), atol=1e-12)
    m = (z > 0.3).double()                      # a fed activation pattern replaces the ReLU's own
    assert torch.allclose(co.cbrn(x, w, b, 1, mask=m), co.instance_norm(z * m), atol=1e-12)


def test_softmax_ce_matches_torch_cross_entropy():
    g = torch.Generator().manual_seed(1)
    logits = torch.randn((500, 18), generator=g, dtype=torch.float64) * 4
    labels = torch.randint(0, 18, (500,), generator=g)
    got = co.softmax_ce(logits, labels)
    ref = F.cross_entropy(logits, labels, reduction="none")
    assert float((got - ref).abs().max()) < 1e-12
    assert float(got.min()) >= 0.0


def _hnm_literal(ce, labels, ratio):
    """model_cutie.py:116-130, statement by statement, with the [Nfg,1] + [1,num_bg] broadcast kept"""
    labels = labels.reshape(-1)
    ce = ce.reshape(-1)
    fg_idx = torch.nonzero(labels != 0)                 # tf.where -> [Nfg, 1]
    fgs = ce[fg_idx]                                    # tf.gather with [Nfg,1] indices -> [Nfg, 1]
    bg_idx = torch.nonzero(labels == 0)
    bgs = ce[bg_idx]                                    # [Nbg, 1]
    num = ratio * fgs.shape[0]
    num_bg = bgs.shape[0] if bgs.shape[0] < num else num
    sorted_bgs = torch.topk(bgs.t(), num_bg, sorted=True).values     # [1, num_bg]
    cross_entropy = fgs + sorted_bgs                    # broadcast [Nfg, num_bg]
    return cross_entropy.mean() if cross_entropy.numel() else ce.new_tensor(float("nan"))


@pytest.mark.parametrize("n,fg_share", [(4096, 0.05), (300, 0.5), (64, 0.9)])
def test_hnm_loss_equals_the_literal_broadcast(n, fg_share):
    g = torch.Generator().manual_seed(n)
    ce = torch.rand(n, generator=g, dtype=torch.float64) * 5
    labels = torch.zeros(n, dtype=torch.int64)
    fg = torch.rand(n, generator=g) < fg_share
    labels[fg] = torch.randint(1, 9, (int(fg.sum()),), generator=g)
    loss, w, n_fg, num_bg = co.hnm_loss(ce, labels, 3)
    lit = _hnm_literal(ce, labels, 3)
    assert abs(float(loss) - float(lit)) < 1e-12
    assert n_fg == int(fg.sum()) and num_bg == min(n - n_fg, 3 * n_fg)
    assert abs(float((w * ce).sum()) - float(loss)) < 1e-12          # the weights are the loss's gradient wrt ce
    assert int((w[labels == 0] != 0).sum()) == num_bg
    if num_bg < n - n_fg:                                            # selected = the largest background CEs
        bg_ce = ce[labels == 0]
        assert float(bg_ce[w[labels == 0] != 0].min()) >= float(bg_ce[w[labels == 0] == 0].max())


def test_hnm_ties_take_the_lower_index_and_empty_sets_are_nan():
    ce = torch.tensor([1.0, 2.0, 2.0, 2.0, 0.5, 2.0], dtype=torch.float64)
    labels = torch.tensor([1, 0, 0, 0, 0, 0])
    loss, w, n_fg, num_bg = co.hnm_loss(ce, labels, 3)       # 3 of the four background cells equal to 2.0
    assert num_bg == 3 and w.tolist() == [1.0, 1 / 3, 1 / 3, 1 / 3, 0.0, 0.0]     # tf.nn.top_k: lower index first
    assert float(loss) == 3.0
    loss, w, n_fg, num_bg = co.hnm_loss(ce, torch.zeros(6, dtype=torch.int64), 3)   # Nfg = 0: mean of an empty tensor
    assert math.isnan(float(loss)) and n_fg == 0 and float(w.abs().sum()) == 0.0
    assert math.isnan(float(_hnm_literal(ce, torch.zeros(6, dtype=torch.int64), 3)))
    loss, _, _, num_bg = co.hnm_loss(ce, torch.ones(6, dtype=torch.int64), 3)       # no background at all
    assert math.isnan(float(loss)) and num_bg == 0


def test_build_loss_l2_covers_conv_kernels_only():
    params = co.init_params(50, 8, 5, seed=0)
    grid = torch.randint(0, 50, (1, 6, 7))
    labels = torch.randint(0, 5, (1, 6, 7))
    L = co.aspp_forward(params, grid)
    ml, reg, tot, logits, sm = co.build_loss(params, L, labels, weight_decay=5e-4)
    want = sum(5e-4 * 0.5 * float((p ** 2).sum()) for k, p in params.items()
               if k.endswith("/weights") and not k.startswith("embedding"))
    assert abs(float(reg) - want) < 1e-12 and abs(float(tot) - float(ml) - want) < 1e-12
    assert logits is L["cls_logits"] and torch.allclose(sm.sum(-1), torch.ones_like(sm.sum(-1)))
    assert list(L["aspp_concat"].shape) == [1, 6, 7, 1024] and list(L["concat1"].shape) == [1, 6, 7, 512]
    assert torch.equal(L["concat1"][..., :256], L["encoder1_1"])          # concat order, model_cutie_aspp.py:67


def test_adam_step_tf_matches_torch_adam_reparameterised():
    """TF: p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps).  torch.optim.Adam divides sqrt(v) by sqrt(1-b2^t)
    BEFORE adding eps, so it equals TF's form when its eps is eps_tf / sqrt(1 - b2^t) at step t."""
    g = torch.Generator().manual_seed(2)
    p0 = {"a/weights": torch.randn((3, 4), generator=g, dtype=torch.float64),
          "a/biases": torch.randn(4, generator=g, dtype=torch.float64)}
    tp = [p0["a/weights"].clone().requires_grad_(True), p0["a/biases"].clone().requires_grad_(True)]
    opt = torch.optim.Adam(tp, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    state, params = {}, p0
    for t in range(1, 6):
        grads = {k: torch.randn(v.shape, generator=g, dtype=torch.float64) * 10 ** (t - 3) for k, v in params.items()}
        gnorm = math.sqrt(sum(float((v ** 2).sum()) for v in grads.values()))
        scale = 10.0 / max(gnorm, 10.0)                  # tf.clip_by_global_norm(grads, 10.0)
        params = co.adam_step_tf(params, grads, state, 1e-2, gnorm)
        for q, k in zip(tp, ("a/weights", "a/biases")):
            q.grad = grads[k] * scale
        opt.param_groups[0]["eps"] = 1e-8 / math.sqrt(1 - 0.999 ** t)
        opt.step()
        for q, k in zip(tp, ("a/weights", "a/biases")):
            assert float((q.detach() - params[k]).abs().max()) < 1e-12, (t, k)


def test_global_norm_uses_undeduplicated_embedding_rows():
    """tf.clip_by_global_norm sees the IndexedSlices values - one row per grid cell - not the summed table gradient"""
    params = co.init_params(30, 8, 5, seed=1)
    grid = torch.zeros((1, 5, 6), dtype=torch.int64)
    grid[0, 0, :3] = 7                                    # three cells on one id: rows are NOT summed before the norm
    labels = torch.zeros((1, 5, 6), dtype=torch.int64)
    labels[0, 0, 0] = 2
    o = co.loss_and_grads(params, grid, labels)
    tf_norm = float(co.global_norm_tf(o["grads"], o["d_embedding_out"]))
    rest = sum(float((gr ** 2).sum()) for k, gr in o["grads"].items() if k != "embedding/weights")
    assert abs(tf_norm ** 2 - rest - float((o["d_embedding_out"] ** 2).sum())) < 1e-12 * tf_norm ** 2
    # the table gradient is the scatter-add of those rows
    dense = torch.zeros_like(params["embedding/weights"])
    dense.index_add_(0, grid.reshape(-1), o["d_embedding_out"].reshape(-1, 8))
    assert float((dense - o["grads"]["embedding/weights"]).abs().max()) < 1e-12


def test_autograd_gradients_match_finite_differences():
    """the backward the HIP kernels are compared with: central differences through the whole small graph"""
    params = co.init_params(20, 4, 3, seed=3)
    # shrink the 256-wide layers' fan-in noise: scale weights up so that activations are O(1)
    grid = torch.randint(0, 20, (1, 5, 6), generator=torch.Generator().manual_seed(4))
    labels = torch.zeros((1, 5, 6), dtype=torch.int64)
    labels[0, 1, 1], labels[0, 3, 4] = 1, 2
    w_cell = torch.rand(30, dtype=torch.float64, generator=torch.Generator().manual_seed(5))
    o = co.loss_and_grads(params, grid, labels, cell_weights=w_cell)
    masks = {k: (o["layers"][k] > o["layers"][k].amin(dim=(1, 2), keepdim=True)).double() for k in co.ASPP_LAYERS}

    def total(ps):
        L = co.aspp_forward(ps, grid, masks=masks)
        return float(co.build_loss(ps, L, labels, cell_weights=w_cell)[2])
    for name, idx in (("cls_logits/weights", (0, 0, 5, 1)), ("decoder1_1/biases", (7,)), ("encoder1_1/weights", (1, 2, 1, 3))):
        h = 1e-6
        pp = {k: v.clone() for k, v in params.items()}
        pm = {k: v.clone() for k, v in params.items()}
        pp[name][idx] += h
        pm[name][idx] -= h
        fd = (total(pp) - total(pm)) / (2 * h)
        om = co.loss_and_grads(params, grid, labels, masks=masks, cell_weights=w_cell)
        an = float(om["grads"][name][idx])
        assert abs(fd - an) < 1e-5 * max(1.0, abs(an)) + 1e-7, (name, fd, an)
