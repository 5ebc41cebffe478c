#!/usr/bin/env python3
"""Soak run (dev tool): N train steps of CUTIE-ASPP in f16 over a changing set of grid shapes and batch sizes, as the
reference's loader produces them (a new rows x cols for almost every batch).  Checks every 25 steps that no conv-tile
meeting timed out, no input counter fired, the loss is finite, and reports loss-scale movements and memory.
usage: python tools/soak.py [steps]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd.model_cutie_aspp import CUTIERes  # noqa: E402
from cutie_amd.runtime import Engine  # noqa: E402
from cutie_amd.train import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
V, E, C = 20000, 128, 18
net = CUTIERes(V, C, types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=32))
net.build_loss()
eng = Engine(net, "cuda:0", dtype="f16", seed=0)
tr = Trainer(eng, learning_rate=1e-4, scale_growth_interval=200)
shapes = [(32, 64, 64), (32, 80, 48), (16, 48, 96), (32, 64, 64), (8, 128, 64), (32, 56, 72), (5, 64, 64), (32, 64, 64), (24, 40, 40)]
g = torch.Generator().manual_seed(5)
batches = []
for B, H, W in shapes:
    occ = torch.rand((B, H, W), generator=g) < 0.12
    grid = torch.where(occ, torch.randint(1, V - 1, (B, H, W), generator=g), torch.zeros((), dtype=torch.int64)).int().cuda()
    labels = torch.where(occ & (torch.rand((B, H, W), generator=g) < 0.25), torch.randint(1, C, (B, H, W), generator=g),
                         torch.zeros((), dtype=torch.int64)).int().cuda()
    batches.append({"grid_table": grid, "gt_classes": labels})
t0 = time.time()
first = last = None
for it in range(steps):
    b = batches[it % len(batches)]
    fetch = it % 25 == 0 or it == steps - 1
    out = tr.step(b, fetch=fetch)
    if fetch:
        P = eng.forward(b["grid_table"], labels=b["gt_classes"])
        eng.check_inputs(P)                         # raises on a timed-out tile meeting or a fired input counter
        for Q in list(eng._plans.values()) if hasattr(eng, "_plans") else []:
            eng.check_inputs(Q)
        loss = out["total_loss"]
        assert loss == loss and abs(loss) < 1e6, "loss %r at step %d" % (loss, it)
        first = loss if first is None else first
        last = loss
        s, skipped = eng.loss_scale_value()
        print("step %5d  shape %s  total_loss %.4f  loss scale %g  skipped %d  mem %.2f GB  %.1f steps/s" % (
            it, tuple(b["grid_table"].shape), loss, s, skipped, torch.cuda.memory_allocated() / 2 ** 30, (it + 1) / (time.time() - t0)),
            flush=True)
print("SOAK OK: %d steps, loss %.4f -> %.4f" % (steps, first, last))
