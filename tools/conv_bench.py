#!/usr/bin/env python3
"""Single-layer timing of the conv kernels (dev tool; not part of the product or the tests).
usage: python tools/conv_bench.py [fwd|wgrad] [dil] [iters] [cin] [cout]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cutie_amd import ops, capi

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dil = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cin = int(sys.argv[4]) if len(sys.argv) > 4 else 256
cout = int(sys.argv[5]) if len(sys.argv) > 5 else 256
B, H, W, kh, kw = int(os.environ.get("CB_B", 32)), 64, 64, 3, 5
dev = "cuda:0"
TDT, CDT = (torch.bfloat16, capi.BF16) if os.environ.get("CB_DTYPE", "f16") == "bf16" else (torch.float16, capi.F16)
g = torch.Generator().manual_seed(0)
x = torch.randn((B, H, W, cin), generator=g)
if os.environ.get("CB_DATA", "random") == "real":
    # what a conv of the step reads: a ReLU'd activation, instance-normalised per (sample, channel) - half of the elements of a
    # plane hold one value (the operands' toggling sets the clock the chip holds)
    x = torch.relu(x)
    x = (x - x.mean(dim=(1, 2), keepdim=True)) / x.std(dim=(1, 2), keepdim=True)
x = x.to(TDT).to(dev)
dz = (torch.randn((B, H, W, cout), generator=g)).to(TDT).to(dev)
w = (torch.randn((kh, kw, cin, cout), generator=g) * (0.01 if os.environ.get("CB_DATA", "random") == "real" else 0.05)).to(dev)
wf, wd = ops.pack_weights(w, CDT)
out = torch.zeros((B, H, W, cout), dtype=TDT, device=dev)
bias = torch.zeros(cout, device=dev)
stats = torch.zeros((B, ops.conv_stats_tiles(H, W), cout, 2), device=dev)
dw = torch.zeros((kh, kw, cin, cout), device=dev)
ws = torch.empty(ops.conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw), dtype=torch.uint8, device=dev)


def run():
    if mode == "fwd":
        ops.conv_fwd([dict(x=x, w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out, cout, bias=bias, relu=True, stats=stats)
    else:
        ops.conv_wgrad(x, cin, dz, cout, kh, kw, dil, dw, ws)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
e0.record()
t0 = time.perf_counter()
for _ in range(iters):
    run()
host_us = (time.perf_counter() - t0) / iters * 1e6
e1.record()
torch.cuda.synchronize()
print("host launch time per call: %.1f us" % host_us)
ms = e0.elapsed_time(e1) / iters
hs = sum(max(0, H - abs((i - 1) * dil)) for i in range(3))
wsum = sum(max(0, W - abs((j - 2) * dil)) for j in range(5))
fl = 2 * B * hs * wsum * cin * cout
print("%s dil=%d cin=%d cout=%d: %.1f us  %.1f TFLOP/s in-bounds (%.1f nominal)" %
      (mode, dil, cin, cout, ms * 1e3, fl / ms / 1e9, 2 * B * H * W * 15 * cin * cout / ms / 1e9))

