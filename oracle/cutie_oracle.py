"""ORACLE (test infrastructure, not product code): CPU restatement of the CUTIE model path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (cutie_amd.*) never does and fails loudly without its HIP library.

PARITY UNPINNED against TensorFlow: the reference's arithmetic lives in tensorflow==1.12.0
(/root/reference/requirements.txt:35), which is not installed here and has no Python 3.10
build; the reference ships no tests, checkpoints or golden activations.
This restatement follows the op definitions at the reference's call sites, on PyTorch's CPU
kernels (an implementation independent of the HIP kernels it checks), in float64 by default:

  embed            model_framework.py:43-54    table[ids]; dropout keep_prob=1 is the identity
  conv/dilate_conv model_framework.py:168-204  stride-1 SAME cross-correlation, HWIO filters,
                                               taps at ((i-kh//2)*rate, (j-kw//2)*rate)
  convolution      model_framework.py:414-424  IN(act(conv(x, W) + b)), act = relu | sigmoid
  instance_norm    tf.contrib.layers.instance_norm(center=False, scale=False): biased variance
                   over H,W per (sample, channel), epsilon 1e-6 (TF 1.12 default)
  concat           model_framework.py:294-296  channel axis
  global_pool      model_framework.py:309-315  reduce_mean over H,W + nearest-neighbour resize
  softmax          model_framework.py:318-320
  graph            model_cutie_aspp.py:35-74   (CUTIERes.setup)
  build_loss       model_cutie.py:107-140      sparse softmax CE, hard negative mining as
                   mean(fgs[Nfg,1] + top_k(bgs)[1,num_bg]) = mean(fg) + mean(top bg), L2
  train step       main_train_json.py:150-158  tf.gradients, clip_by_global_norm(10), Adam
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

IN_EPS = 1e-6


def instance_norm(a, eps=IN_EPS):
    """a: [B,H,W,C]."""
    mean = a.mean(dim=(1, 2), keepdim=True)
    var = ((a - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    return (a - mean) * torch.rsqrt(var + eps)


def conv_raw(x, w_hwio, rate=1):
    """x [B,H,W,Cin], w [kh,kw,Cin,Cout] -> [B,H,W,Cout]; SAME, stride 1, dilation `rate`."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    y = F.conv2d(x.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), None, stride=1,
                 padding=((kh // 2) * rate, (kw // 2) * rate), dilation=rate)
    return y.permute(0, 2, 3, 1)


def cbrn(x, w_hwio, bias, rate=1, activation="relu", mask=None, pre=None, name=None):
    """mask (optional, relu only): a 0/1 tensor that replaces the ReLU's own activation pattern
    (a = z * mask).  Tests pass the DEVICE's pattern so that gradients are compared on the same
    side of every ReLU kink; a pre-activation within rounding of 0 otherwise flips the whole
    local gradient and says nothing about the backward kernels."""
    z = conv_raw(x, w_hwio, rate) + bias
    if pre is not None:             # the pre-activation conv + b: its gradient is what the device calls dz
        if z.requires_grad:
            z.retain_grad()
        pre[name] = z
    if mask is not None and activation == "relu":
        a = z * mask
    else:
        a = torch.relu(z) if activation == "relu" else torch.sigmoid(z)
    return instance_norm(a)


def global_pool(x):
    return x.mean(dim=(1, 2), keepdim=True).expand_as(x)


# layer name -> (kh, kw, cout, rate); cin follows from the graph below
ASPP_LAYERS = OrderedDict([
    ("encoder1_1", (3, 5, 256, 1)), ("encoder1_2", (3, 5, 256, 1)), ("encoder1_3", (3, 5, 256, 1)),
    ("encoder1_4", (3, 5, 256, 1)), ("encoder1_5", (3, 5, 256, 2)), ("encoder1_6", (3, 5, 256, 4)),
    ("encoder1_7", (3, 5, 256, 8)), ("encoder1_8", (3, 5, 256, 16)),
    ("aspp_1", (3, 5, 256, 4)), ("aspp_2", (3, 5, 256, 8)), ("aspp_3", (3, 5, 256, 16)),
    ("aspp_1x1", (1, 1, 256, 1)), ("decoder1_1", (3, 5, 64, 1)), ("cls_logits", (1, 1, None, 1)),
])


def aspp_param_shapes(num_vocabs, embedding_size, num_classes):
    """Variable name -> shape, in the order TF creates them (model_cutie_aspp.py:35-74)."""
    cin = {"encoder1_1": embedding_size, "aspp_1": 256, "aspp_2": 256, "aspp_3": 256,
           "aspp_1x1": 1024, "decoder1_1": 512, "cls_logits": 64}
    shapes = OrderedDict()
    shapes["embedding/weights"] = (num_vocabs, embedding_size)
    prev = None
    for name, (kh, kw, cout, _rate) in ASPP_LAYERS.items():
        c_in = cin.get(name, prev)
        c_out = num_classes if cout is None else cout
        shapes[name + "/weights"] = (kh, kw, c_in, c_out)
        shapes[name + "/biases"] = (c_out,)
        prev = c_out
    return shapes


def init_params(num_vocabs, embedding_size, num_classes, seed=0, dtype=torch.float64):
    """Seeded stand-ins for TF's initialisers (model_framework.py:46,178-179): embedding
    U(-1,1), conv W truncated normal sigma 0.01 (resampled beyond 2 sigma), biases 0."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape in aspp_param_shapes(num_vocabs, embedding_size, num_classes).items():
        if name == "embedding/weights":
            p = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        elif name.endswith("/weights"):
            p = torch.randn(shape, generator=g, dtype=torch.float64)
            for _ in range(8):
                bad = p.abs() > 2
                if not bad.any():
                    break
                p = torch.where(bad, torch.randn(shape, generator=g, dtype=torch.float64), p)
            p = p.clamp(-2, 2) * 0.01
        else:
            p = torch.zeros(shape, dtype=torch.float64)
        params[name] = p.to(dtype)
    return params


def aspp_forward(params, grid, activation="relu", masks=None, pre=None):
    """grid int [B,H,W] or [B,H,W,1] -> dict of every named layer (model_cutie_aspp.py:35-74)."""
    if grid.dim() == 4:
        grid = grid[..., 0]
    L = OrderedDict()
    m = (lambda n: masks.get(n)) if masks else (lambda n: None)
    L["embedding"] = params["embedding/weights"][grid.long()]
    x = L["embedding"]
    for name in ("encoder1_1", "encoder1_2", "encoder1_3", "encoder1_4", "encoder1_5", "encoder1_6",
                 "encoder1_7", "encoder1_8"):
        x = cbrn(x, params[name + "/weights"], params[name + "/biases"], ASPP_LAYERS[name][3], mask=m(name), pre=pre, name=name)
        L[name] = x
    e8 = L["encoder1_8"]
    for name in ("aspp_1", "aspp_2", "aspp_3"):
        L[name] = cbrn(e8, params[name + "/weights"], params[name + "/biases"], ASPP_LAYERS[name][3], mask=m(name),
                       pre=pre, name=name)
    L["aspp_4"] = global_pool(e8)
    L["aspp_concat"] = torch.cat([L["aspp_1"], L["aspp_2"], L["aspp_3"], L["aspp_4"]], dim=3)
    L["aspp_1x1"] = cbrn(L["aspp_concat"], params["aspp_1x1/weights"], params["aspp_1x1/biases"], mask=m("aspp_1x1"),
                           pre=pre, name="aspp_1x1")
    L["concat1"] = torch.cat([L["encoder1_1"], L["aspp_1x1"]], dim=3)
    L["decoder1_1"] = cbrn(L["concat1"], params["decoder1_1/weights"], params["decoder1_1/biases"],
                           mask=m("decoder1_1"), pre=pre, name="decoder1_1")
    L["cls_logits"] = cbrn(L["decoder1_1"], params["cls_logits/weights"], params["cls_logits/biases"],
                           activation=activation, mask=m("cls_logits"), pre=pre, name="cls_logits")
    L["softmax"] = torch.softmax(L["cls_logits"], dim=-1)
    return L


def softmax_ce(logits, labels):
    """per-cell sparse softmax cross-entropy; logits [N,C], labels [N]."""
    return torch.logsumexp(logits, dim=-1) - logits.gather(-1, labels.long().unsqueeze(-1)).squeeze(-1)


def hnm_loss(ce, labels, ratio):
    """model_cutie.py:116-130.  Returns (model_loss, per-cell weight) with
    model_loss == (weight * ce).sum() when both sets are non-empty."""
    ce = ce.reshape(-1)
    labels = labels.reshape(-1)
    fg = labels != 0
    bg = ~fg
    n_fg, n_bg = int(fg.sum()), int(bg.sum())
    num_bg = min(n_bg, int(ratio) * n_fg)
    weight = torch.zeros_like(ce)
    if n_fg == 0 or num_bg == 0:
        return ce.new_tensor(float("nan")), weight, n_fg, num_bg
    bg_idx = torch.nonzero(bg).reshape(-1)
    bg_ce = ce[bg_idx]
    # tf.nn.top_k: largest first, ties -> lower index first (stable sort of -ce)
    order = torch.sort(-bg_ce.detach(), stable=True).indices[:num_bg]
    sel = bg_idx[order]
    weight[fg] = 1.0 / n_fg
    weight[sel] = 1.0 / num_bg
    loss = ce[fg].mean() + ce[sel].mean()
    return loss, weight, n_fg, num_bg


def calc_ghm_weights(logits, labels, bins=30, momentum=0.0, eps=1e-6, acc_sum=None):
    """main_train_json.py:86-119, vectorised but with the same arithmetic: float64 `grad`, the same
    edge list, the same in-place momentum update of `acc_sum` (a list of `bins` floats, updated here).
    logits [B,H,W,C] (the fetched cls_logits), labels [B,H,W] -> float64 weights [B,H,W,C]."""
    import numpy as np
    lg = np.asarray(logits.detach().cpu().numpy() if torch.is_tensor(logits) else logits)
    lb = np.asarray(labels.detach().cpu().numpy() if torch.is_tensor(labels) else labels).reshape(-1).astype(np.int64)
    shape = lg.shape
    C = shape[-1]
    edges = [float(x) / bins for x in range(bins + 1)]
    edges[-1] += eps
    if acc_sum is None:
        acc_sum = [0.0 for _ in range(bins)]
    onehot = np.zeros((lb.size, C), dtype=np.int64)
    onehot[np.arange(lb.size), lb] = 1
    grad = np.abs(lg.reshape(-1).astype(np.float32) - onehot.reshape(-1))        # float32 - int64 -> float64
    weights = np.ones(grad.shape)
    n_total = lg.size
    m = 0
    for i in range(bins):
        idx = (grad >= edges[i]) & (grad < edges[i + 1])
        num = int(idx.sum())
        if num > 0:
            acc_sum[i] = momentum * acc_sum[i] + (1 - momentum) * num
            weights[idx] = n_total / acc_sum[i]
            m += 1
    if m > 0:
        weights = weights / m
    return torch.from_numpy(weights.reshape(shape)), acc_sum


def build_loss(params, layers, labels, hard_negative_ratio=3, weight_decay=5e-4, cell_weights=None, ghm_weights=None):
    """(model_loss, regularization_loss, total_loss, logits, softmax) - model_cutie.py:140.
    cell_weights (optional): a fixed per-cell weight vector replacing the hard-negative
    selection (tests pass the DEVICE's selection so that a top-k boundary that falls between
    two nearly equal CE values does not masquerade as a gradient error)."""
    logits = layers["cls_logits"]
    C = logits.shape[-1]
    if ghm_weights is not None:      # model_cutie.py:110; the weights are a fed constant (float32 placeholder)
        logits = logits * ghm_weights.to(torch.float32).to(logits.dtype)
    ce = softmax_ce(logits.reshape(-1, C), labels.reshape(-1))
    if cell_weights is not None:
        model_loss = (cell_weights.reshape(-1).to(ce.dtype) * ce).sum()
    else:
        model_loss, _w, _nfg, _nbg = hnm_loss(ce, labels, hard_negative_ratio)
    reg = sum(weight_decay * 0.5 * (p ** 2).sum() for n, p in params.items()
              if n.endswith("/weights") and n != "embedding/weights")
    return model_loss, reg, model_loss + reg, layers["cls_logits"], layers["softmax"]


def loss_and_grads(params, grid, labels, hard_negative_ratio=3, weight_decay=5e-4, masks=None, cell_weights=None,
                   activation="relu", ghm_weights=None):
    """activation='sigmoid' + ghm_weights: the use_ghm = 1 graph (model_cutie_aspp.py:17,73; model_cutie.py:110)"""
    ps = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    pre = OrderedDict()
    layers = aspp_forward(ps, grid, activation=activation, masks=masks, pre=pre)
    for t in layers.values():
        if t.requires_grad:
            t.retain_grad()
    model_loss, reg, total, logits, sm = build_loss(ps, layers, labels, hard_negative_ratio, weight_decay,
                                                    cell_weights, ghm_weights)
    total.backward()
    grads = OrderedDict((k, v.grad) for k, v in ps.items())
    return {"model_loss": model_loss.detach(), "reg_loss": reg.detach(), "total_loss": total.detach(),
            "layers": layers, "grads": grads, "d_embedding_out": layers["embedding"].grad,
            "dlayers": OrderedDict((k, t.grad) for k, t in layers.items() if t.grad is not None),
            "dpre": OrderedDict((k, t.grad) for k, t in pre.items() if t.grad is not None)}


def global_norm_tf(grads, d_embedding_out):
    """tf.clip_by_global_norm over tf.gradients' output: the embedding gradient is an
    IndexedSlices whose `values` (one row per grid cell, NOT summed per id) enter the norm."""
    sq = 0.0
    for k, g in grads.items():
        if k == "embedding/weights":
            sq = sq + (d_embedding_out ** 2).sum()
        else:
            sq = sq + (g ** 2).sum()
    return torch.sqrt(sq)


def adam_step_tf(params, grads, state, lr, gnorm, clip=10.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (TF 1.12): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    p -= lr_t * m / (sqrt(v) + eps).  Sparse apply on the embedding equals the dense form."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    scale = clip / max(float(gnorm), clip)
    out = OrderedDict()
    for k, p in params.items():
        g = grads[k] * scale
        m = state.setdefault("m", {}).get(k, torch.zeros_like(p))
        v = state.setdefault("v", {}).get(k, torch.zeros_like(p))
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        state["m"][k], state["v"][k] = m, v
        out[k] = p - lr_t * m / (v.sqrt() + eps)
    return out
