#!/usr/bin/env python3
"""Host-side issue time of one train step vs. its GPU time (dev tool)."""
import os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops
from cutie_amd.model_cutie_aspp import CUTIERes
from cutie_amd.runtime import Engine
from cutie_amd.synthetic import make_documents
from cutie_amd.train import Trainer

dev = "cuda:0"
V, E, C, H, W, B = 20000, 128, 18, 64, 64, 32
p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
net = CUTIERes(V, C, p); net.build_loss()
eng = Engine(net, dev, dtype="f16", seed=0)
tr = Trainer(eng)
tok = {k: torch.from_numpy(v).to(dev) for k, v in make_documents(B, C, seed=1234).items()}
def step():
    g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], H, W, 1)
    tr.step({"grid_table": g["grid"], "gt_classes": g["labels"]}, fetch=False)
for _ in range(5): step()
torch.cuda.synchronize()
for n in (1, 2, 4):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %d: host issue %.2f ms/step, total %.2f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
