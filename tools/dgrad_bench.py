#!/usr/bin/env python3
"""Data gradient of an instance-normalised 256 -> 256 layer at the bench shape, three ways (dev tool):
  a) cutie_conv_dgrad_in_stats (data gradient + fused sums) followed by cutie_in_bwd_apply   (two-call form)
  b) cutie_conv_dgrad_in_bwd with a sync area                                                (one launch, tiles meet)
usage: python tools/dgrad_bench.py [dil] [iters] [KHxKW]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops, capi

dil = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
KH, KW = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "3x5").split("x"))      # 1x1: the ASPP fan-out data gradients
B, H, W, C = 32, 64, 64, 256
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tdt = torch.float16
dzin = (torch.randn((B, H, W, C), generator=g) * 0.7).to(tdt).to(dev)
a = torch.relu(torch.randn((B, H, W, C), generator=g) + 0.2).to(tdt).to(dev)
w = (torch.randn((KH, KW, C, C), generator=g) * 0.05).to(dev)
_, wd = ops.pack_weights(w, capi.F16, want_fwd=False)
T = ops.conv_stats_tiles(H, W)
stats = torch.zeros((B, T, C, 2), device=dev)
ops.in_stats(a, stats)
red = torch.zeros((B, T, C, 2), device=dev)
dbp = torch.zeros((B * 32, C), device=dev)
dy = torch.zeros((B, H, W, C), dtype=tdt, device=dev)
dz = torch.zeros((B, H, W, C), dtype=tdt, device=dev)
sync = torch.zeros(ops.conv_sync_bytes(B, T, C) // 4, dtype=torch.int32, device=dev)
seg = [dict(x=dzin, w=wd, cin=C, kh=KH, kw=KW, dil=dil, flip=True)]


def two_call():
    ops.conv_fwd(seg, dy, C, norm_bwd=(a, stats, red))
    ops.in_bwd_apply(dy, a, stats, red, dz, dbias_part=dbp)


def dgrad_only():
    ops.conv_fwd(seg, dy, C, norm_bwd=(a, stats, red))


def one_launch():
    ops.conv_dgrad_in_bwd(seg, dy, C, a, stats, red, dz, dbias_part=dbp, sync=sync)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for r in range(2):
    print("dil %d: dgrad+sums %.1f us | two-call %.1f us | one launch %.1f us (timeouts %d)" % (
        dil, timeit(dgrad_only), timeit(two_call), timeit(one_launch), int(sync[0])))
