#!/usr/bin/env python3
"""Timing of the instance-norm passes on one 256-channel layer (dev tool).  Four sets of tensors are
used in turn (1 GB in all) so that the 256 MB Infinity Cache does not serve the reads; T = slots per
(sample, channel) of the partial-sum arrays."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops

B, H, W, C = 32, 64, 64, 256
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
NS = 4
a = [torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev) for _ in range(NS)]
dy = [torch.randn((B, H, W, C), generator=g).to(torch.bfloat16).to(dev) for _ in range(NS)]
y = [torch.empty_like(a[0]) for _ in range(NS)]


def timeit(name, fn, nbytes, iters=40):
    for k in range(4):
        fn(k % NS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k % NS)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-22s %7.1f us  %5.2f TB/s" % (name, ms * 1e3, nbytes / ms / 1e9), flush=True)


n = a[0].numel() * 2
for T in (1, 16):
    stats = torch.zeros((B, T, C, 2), device=dev)
    red = torch.zeros((B, T, C, 2), device=dev)
    dbp = torch.zeros((B * ops.in_tiles(B, H * W, C), C), device=dev)
    ops.in_stats(a[0], stats)
    timeit("in_stats T=%d" % T, lambda k: ops.in_stats(a[k], stats), n)
    timeit("in_apply T=%d" % T, lambda k: ops.in_apply(a[k], stats, y[k]), 2 * n)
    timeit("in_bwd_stats T=%d" % T, lambda k: ops.in_bwd_stats(dy[k], a[k], stats, red), 2 * n)
    timeit("in_bwd_apply T=%d" % T, lambda k: ops.in_bwd_apply(dy[k], a[k], stats, red, y[k], dbias_part=dbp, relu=True), 3 * n)
