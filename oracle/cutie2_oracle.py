"""ORACLE (test infrastructure, not product code): CPU restatement of the CUTIE2 graphs (forward, and
loss + gradients through autograd).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED against TensorFlow, for the reasons given in oracle/cutie_oracle.py; the ops are
the ones defined there plus

  positional_sampling  model_framework.py:73-86: the image feature map [B,Hi,W,C] is flattened
                       to [B, Hi*W, C] and gathered per sample at ps_1d_indices [B, rows*cols]
                       (tf.batch_gather), then reshaped to the grid's [B,rows,cols,C].

The two graphs are restated as tables from
  model_cutie2_aspp.py:36-107    (CUTIE2 "aspp": 32-channel 3x3 image branch with its own ASPP)
  model_cutie2_dilate.py:36-93   (CUTIE2 "dilate": 128-channel 3x3 image branch, rates 2,4,8,16)
followed by the trunk both share with CUTIE-ASPP (encoder1_1..8, aspp_1..4, aspp_1x1, decoder1_1,
cls_logits, softmax).  Variable creation order = table order (TF creates them as setup() runs).
"""
from collections import OrderedDict

import torch

from .cutie_oracle import build_loss, cbrn, global_pool

# (op, name, inputs, args);  conv args = (kh, kw, cout, rate)
TRUNK = [
    ("conv", "encoder1_1", ["feature_fuser"], (3, 5, 256, 1)),
    ("conv", "encoder1_2", ["encoder1_1"], (3, 5, 256, 1)),
    ("conv", "encoder1_3", ["encoder1_2"], (3, 5, 256, 1)),
    ("conv", "encoder1_4", ["encoder1_3"], (3, 5, 256, 1)),
    ("conv", "encoder1_5", ["encoder1_4"], (3, 5, 256, 2)),
    ("conv", "encoder1_6", ["encoder1_5"], (3, 5, 256, 4)),
    ("conv", "encoder1_7", ["encoder1_6"], (3, 5, 256, 8)),
    ("conv", "encoder1_8", ["encoder1_7"], (3, 5, 256, 16)),
    ("conv", "aspp_1", ["encoder1_8"], (3, 5, 256, 4)),
    ("conv", "aspp_2", ["encoder1_8"], (3, 5, 256, 8)),
    ("conv", "aspp_3", ["encoder1_8"], (3, 5, 256, 16)),
    ("global_pool", "aspp_4", ["encoder1_8"], None),
    ("concat", "aspp_concat", ["aspp_1", "aspp_2", "aspp_3", "aspp_4"], None),
    ("conv", "aspp_1x1", ["aspp_concat"], (1, 1, 256, 1)),
    ("concat", "concat1", ["encoder1_1", "aspp_1x1"], None),
    ("conv", "decoder1_1", ["concat1"], (3, 5, 64, 1)),
    ("conv", "cls_logits", ["decoder1_1"], (1, 1, None, 1)),
    ("softmax", "softmax", ["cls_logits"], None),
]

CUTIE2_ASPP = [
    ("embed", "embedding", ["data_grid"], None),
    ("conv", "image_encoder1_1", ["data_image"], (3, 3, 32, 1)),
    ("conv", "image_encoder1_2", ["image_encoder1_1"], (3, 3, 32, 1)),
    ("conv", "image_encoder1_5", ["image_encoder1_2"], (3, 3, 32, 2)),
    ("conv", "image_encoder1_6", ["image_encoder1_5"], (3, 3, 32, 2)),
    ("conv", "image_encoder1_7", ["image_encoder1_6"], (3, 3, 32, 2)),
    ("conv", "image_encoder1_8", ["image_encoder1_7"], (3, 3, 32, 2)),
    ("conv", "image_aspp_1", ["image_encoder1_8"], (3, 3, 32, 4)),
    ("conv", "image_aspp_2", ["image_encoder1_8"], (3, 3, 32, 8)),
    ("conv", "image_aspp_3", ["image_encoder1_8"], (3, 3, 32, 16)),
    ("global_pool", "image_aspp_4", ["image_encoder1_8"], None),
    ("concat", "image_aspp_concat", ["image_aspp_1", "image_aspp_2", "image_aspp_3", "image_aspp_4"], None),
    ("conv", "image_aspp_1x1", ["image_aspp_concat"], (1, 1, 32, 1)),
    ("concat", "image_concat1", ["image_encoder1_1", "image_aspp_1x1"], None),
    ("conv", "image_featuremap", ["image_concat1"], (3, 3, 32, 1)),
    ("positional_sampling", "positional_sampling", ["image_featuremap", "ps_1d_indices", "data_grid"], 32),
    ("concat", "concat", ["positional_sampling", "embedding"], None),
    ("conv", "feature_fuser", ["concat"], (1, 1, 256, 1)),
] + TRUNK

CUTIE2_DILATE = [
    ("embed", "embedding", ["data_grid"], None),
    ("conv", "image_encoder1_1", ["data_image"], (3, 3, 128, 1)),
    ("conv", "image_encoder1_2", ["image_encoder1_1"], (3, 3, 128, 1)),
    ("conv", "image_encoder1_5", ["image_encoder1_2"], (3, 3, 128, 2)),
    ("conv", "image_encoder1_6", ["image_encoder1_5"], (3, 3, 128, 4)),
    ("conv", "image_encoder1_7", ["image_encoder1_6"], (3, 3, 128, 8)),
    ("conv", "image_encoder1_8", ["image_encoder1_7"], (3, 3, 128, 16)),
    ("positional_sampling", "positional_sampling", ["image_encoder1_8", "ps_1d_indices", "data_grid"], 128),
    ("concat", "concat", ["positional_sampling", "embedding"], None),
    ("conv", "feature_fuser", ["concat"], (1, 1, 256, 1)),
] + TRUNK

GRAPHS = {"aspp": CUTIE2_ASPP, "dilate": CUTIE2_DILATE}


def positional_sampling(featuremap, ps_1d_indices, grid_shape, c):
    """model_framework.py:73-86"""
    b, h, w = grid_shape
    flat = featuremap.reshape(b, -1, c)
    idx = ps_1d_indices.reshape(b, -1).long()
    out = torch.gather(flat, 1, idx.unsqueeze(-1).expand(-1, -1, c))
    return out.reshape(b, h, w, c)


def param_shapes(graph, num_vocabs, embedding_size, num_classes):
    """Variable name -> shape in creation order; channel counts follow the graph."""
    ch = {"data_image": 3}
    shapes = OrderedDict()
    for op, name, inputs, args in graph:
        if op == "embed":
            shapes[name + "/weights"] = (num_vocabs, embedding_size)
            ch[name] = embedding_size
        elif op == "conv":
            kh, kw, cout, _rate = args
            cout = num_classes if cout is None else cout
            shapes[name + "/weights"] = (kh, kw, ch[inputs[0]], cout)
            shapes[name + "/biases"] = (cout,)
            ch[name] = cout
        elif op == "concat":
            ch[name] = sum(ch[i] for i in inputs)
        elif op == "global_pool":
            ch[name] = ch[inputs[0]]
        elif op == "positional_sampling":
            ch[name] = args
        elif op == "softmax":
            ch[name] = ch[inputs[0]]
    return shapes


def init_params(graph, num_vocabs, embedding_size, num_classes, seed=0, dtype=torch.float64, weight_std=0.01):
    """Seeded stand-ins for TF's initialisers (model_framework.py:46,178-179).  `weight_std` lets a
    test use larger kernels: with sigma 0.01 and 3 input channels the image branch's activations
    are tiny and every layer's instance norm divides by ~eps."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape in param_shapes(graph, num_vocabs, embedding_size, num_classes).items():
        if len(shape) == 2:
            p = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        elif len(shape) == 4:
            p = torch.randn(shape, generator=g, dtype=torch.float64).clamp(-2, 2) * weight_std
        else:
            p = torch.zeros(shape, dtype=torch.float64)
        params[name] = p.to(dtype)
    return params


def forward(graph, params, grid, image, ps_1d_indices, activation="relu", masks=None, pre=None):
    """grid int [B,H,W]; image [B,Hi,W,3]; ps_1d_indices int [B,H*W] -> dict of every named layer.
    masks: optional layer name -> 0/1 ReLU pattern (see cutie_oracle.cbrn)."""
    if grid.dim() == 4:
        grid = grid[..., 0]
    L = OrderedDict(data_grid=grid, data_image=image, ps_1d_indices=ps_1d_indices)
    for op, name, inputs, args in graph:
        if op == "embed":
            L[name] = params[name + "/weights"][grid.long()]
        elif op == "conv":
            act = activation if name == "cls_logits" else "relu"
            L[name] = cbrn(L[inputs[0]], params[name + "/weights"], params[name + "/biases"], args[3], activation=act,
                           mask=masks.get(name) if masks else None, pre=pre, name=name)
        elif op == "concat":
            L[name] = torch.cat([L[i] for i in inputs], dim=3)
        elif op == "global_pool":
            L[name] = global_pool(L[inputs[0]])
        elif op == "positional_sampling":
            L[name] = positional_sampling(L[inputs[0]], L[inputs[1]], tuple(grid.shape), args)
        elif op == "softmax":
            L[name] = torch.softmax(L[inputs[0]], dim=-1)
    return L


def loss_and_grads(graph, params, grid, image, ps_1d_indices, labels, hard_negative_ratio=3, weight_decay=5e-4,
                   masks=None, cell_weights=None):
    """model_cutie2.py:114-147 (the same loss as model_cutie.py:107-140) + tf.gradients of total_loss
    w.r.t. every variable, by autograd over the restated graph."""
    ps = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
    pre = OrderedDict()
    layers = forward(graph, ps, grid, image, ps_1d_indices, masks=masks, pre=pre)
    for t in layers.values():
        if torch.is_tensor(t) and t.requires_grad:
            t.retain_grad()
    model_loss, reg, total, _logits, _sm = build_loss(ps, layers, labels, hard_negative_ratio, weight_decay, cell_weights)
    total.backward()
    return {"model_loss": model_loss.detach(), "reg_loss": reg.detach(), "total_loss": total.detach(), "layers": layers,
            "grads": OrderedDict((k, v.grad) for k, v in ps.items()),
            "dlayers": OrderedDict((k, t.grad) for k, t in layers.items() if torch.is_tensor(t) and t.grad is not None),
            "dpre": OrderedDict((k, t.grad) for k, t in pre.items() if t.grad is not None)}
