#!/usr/bin/env python3
"""Timing of the instance-norm passes on one 256-channel layer (dev tool)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops

B, H, W, C = 32, 64, 64, 256
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
a = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev)
dy = torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev)
y = torch.empty_like(a)
dz = torch.empty_like(a)
stats = torch.zeros((B, C, 2), device=dev)
red = torch.zeros((B, C, 2), device=dev)
dbp = torch.zeros((B, C), device=dev)
ops.in_stats(a, stats)


def timeit(name, fn, nbytes, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-14s %7.1f us  %5.2f TB/s" % (name, ms * 1e3, nbytes / ms / 1e9))


n = a.numel() * 2
timeit("in_stats", lambda: ops.in_stats(a, stats), n)
timeit("in_apply", lambda: ops.in_apply(a, stats, y), 2 * n)
timeit("in_bwd_stats", lambda: ops.in_bwd_stats(dy, a, stats, red), 2 * n)
timeit("in_bwd_apply", lambda: ops.in_bwd_apply(dy, a, stats, red, dz, dbias_part=dbp, relu=True), 3 * n)
