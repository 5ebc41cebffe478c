"""conv_duo_kernel (4-wave workgroups, two a CU, 256 px x 128 ch tiles) against conv_wide_kernel on a 256-channel layer.

The library reads CUTIE_CONV_DUO once when it is loaded, so each setting runs in a process of its own (one after the other:
one process on the GPU at a time).  Both kernels add the K halves of a (64-channel chunk, tap) step in the same order and share
the epilogue with the same wave / lane partition of the tile, so every output - the plain forward, the forward with the norm
inside (y, moments) and the one-launch data gradient + norm gradient (dz, bias-gradient partials) - must agree BIT FOR BIT.
That is what lets CUTIE_CONV_DUO=2 be measured against the shipped path with nothing but the schedule changed (DESIGN.md
section 3, "Two workgroups a CU"); the oracle parity of both is the business of tests/test_kernels_gpu.py."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
import torch
sys.path.insert(0, %r)
from cutie_amd import ops, capi
dev = "cuda:0"
B, H, W, C = 8, 64, 64, 256
g = torch.Generator().manual_seed(3)
out = {}
def sha(t):
    return hashlib.sha256(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()
for dil in (1, 16):
    x = torch.relu(torch.randn((B, H, W, C), generator=g)).to(torch.float16).to(dev)
    w = (torch.randn((3, 5, C, C), generator=g) * 0.03).to(dev)
    bias = (torch.randn(C, generator=g) * 0.1).to(dev)
    wf, wd = ops.pack_weights(w, capi.F16)
    T = ops.conv_stats_tiles(H, W)
    # forward with the norm inside (or conv + in_apply where the tiles are too unbalanced to meet)
    a = torch.zeros((B, H, W, C), dtype=torch.float16, device=dev)
    y = torch.zeros((B, H, W, C), dtype=torch.float16, device=dev)
    stats = torch.zeros((B, T, C, 2), device=dev)
    sync = torch.zeros(ops.conv_sync_bytes(B, T, C) // 4, dtype=torch.int32, device=dev)
    late = torch.zeros(1, dtype=torch.int32, device=dev)
    seg = [dict(x=x, w=wf, cin=C, kh=3, kw=5, dil=dil)]
    fused = ops.conv_in_fwd(seg, a, y, C, bias, "relu", stats, sync=sync, late=late)
    if not fused:
        ops.in_apply(a, stats, y, bias=bias, relu="relu")
    out["fwd_y_d%%d" %% dil] = sha(y)
    out["fwd_stats_d%%d" %% dil] = sha(stats)
    # plain forward (bias + ReLU, moments)
    a2 = torch.zeros_like(a)
    st2 = torch.zeros_like(stats)
    ops.conv_fwd(seg, a2, C, bias=bias, relu="relu", stats=st2)
    out["plain_a_d%%d" %% dil] = sha(a2)
    out["plain_stats_d%%d" %% dil] = sha(st2)
    # data gradient + the whole norm gradient of the layer that produced y
    dzin = (torch.randn((B, H, W, C), generator=g) * 0.5).to(torch.float16).to(dev)
    dy = torch.zeros_like(a)
    dz = torch.zeros_like(a)
    red = torch.zeros((B, T, C, 2), device=dev)
    dbp = torch.zeros((B * 32, C), device=dev)
    sync2 = torch.zeros(ops.conv_sync_bytes(B, T, C) // 4, dtype=torch.int32, device=dev)
    av, from_y = (y, True) if fused else (a, False)
    ops.conv_dgrad_in_bwd([dict(x=dzin, w=wd, cin=C, kh=3, kw=5, dil=dil, flip=True)], dy, C, av, stats, red, dz,
                          dbias_part=dbp, relu="relu", sync=sync2, bias=bias, from_y=from_y, late=late)
    out["bwd_dz_d%%d" %% dil] = sha(dz)
    if dil == 1:      # (at dilation 16 the two forms may split the call differently: other partial rows, the same dz)
        out["bwd_dbias_d%%d" %% dil] = sha(dbp)
    out["fused_d%%d" %% dil] = bool(fused)
torch.cuda.synchronize()
assert int(late) == 0
print("RESULT " + json.dumps(out, sort_keys=True))
"""


def _run(duo):
    env = dict(os.environ, CUTIE_CONV_DUO=str(duo))
    r = subprocess.run([sys.executable, "-c", CHILD % REPO], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_duo_kernel_produces_the_bits_of_the_8_wave_kernel_on_a_256_channel_layer():
    base = _run(0)
    duo = _run(2)
    assert base["fused_d1"] and duo["fused_d1"]      # dilation 1: both forms take the one-launch forward
    diff = [k for k in base if base[k] != duo[k]]
    assert not diff, diff
