#!/usr/bin/env python3
"""Timing of cutie_grid_build on the bench batch (dev tool)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops
from cutie_amd.synthetic import make_documents

dev = "cuda:0"
for B in (32, 256):
    packed = make_documents(B, 18, seed=1234)
    tok = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
    f = lambda: ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], 64, 64, 1)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("B=%d: %.1f us per call, %.0f docs/s" % (B, ms * 1e3, B / ms * 1e3), flush=True)
