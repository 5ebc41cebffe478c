#!/usr/bin/env python3
"""Inference throughput of CUTIE-ASPP 64x64 (BASELINE.json configs[1]'s network, forward only) - dev tool for the
A/B of the one-launch conv + instance norm (CUTIE_RENDEZVOUS=0/1).  usage: python tools/infer_bench.py [B] [steps]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd.model_cutie_aspp import CUTIERes  # noqa: E402
from cutie_amd.runtime import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
V, E, C, H, W = 20000, 128, 18, 64, 64
net = CUTIERes(V, C, types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B))
net.build_loss()
eng = Engine(net, "cuda:0", dtype="f16", seed=0)
g = torch.Generator().manual_seed(1)
occ = torch.rand((B, H, W), generator=g) < 0.12
grid = torch.where(occ, torch.randint(1, V - 1, (B, H, W), generator=g), torch.zeros((), dtype=torch.int64)).int().cuda()
for _ in range(15):
    P = eng.forward(grid)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    P = eng.forward(grid)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
eng.check_inputs(P)
print("CUTIE-ASPP 64x64 inference B=%d f16: %.3f ms/step, %.0f docs/s (%.1f %% of the bf16 MFMA peak, 77.248 GF/doc in-bounds)"
      % (B, 1e3 * dt, B / dt, 100 * B / dt * 77.248e9 / 2516.6e12))
