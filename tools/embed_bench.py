#!/usr/bin/env python3
"""What bounds embed_bwd (dev tool): the bench batch's ids, all-zero ids, all-distinct ids, one hot id."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops
from cutie_amd.synthetic import make_documents

dev = "cuda:0"
V, E, n = 20000, 128, 32 * 64 * 64
packed = make_documents(32, 18, seed=1234)
tok = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], 64, 64, 1)
real = g["grid"].reshape(-1).contiguous()
gen = torch.Generator().manual_seed(0)
dout = torch.randn((n, E), generator=gen).to(torch.bfloat16).to(dev)
dtab = torch.zeros((V, E), device=dev)
ws = torch.empty(ops.embed_bwd_workspace_bytes(n, E), dtype=torch.uint8, device=dev)
cases = {
    "bench ids": real,
    "all zero": torch.zeros_like(real),
    "distinct (cell % V)": (torch.arange(n, device=dev, dtype=torch.int32) % (V - 2) + 2),
    "bench, ids>=2 made distinct": torch.where(real >= 2, torch.arange(n, device=dev, dtype=torch.int32) % (V - 2) + 2, real),
    "bench, ids>=2 all = 2": torch.where(real >= 2, torch.full_like(real, 2), real),
}
for name, ids in cases.items():
    for _ in range(3):
        ops.embed_bwd(ids, dout, dtab, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.embed_bwd(ids, dout, dtab, ws)
    e1.record()
    torch.cuda.synchronize()
    print("%-32s %7.1f us   cells with id >= 2: %d" % (name, e0.elapsed_time(e1) / 20 * 1e3, int((ids >= 2).sum())), flush=True)
