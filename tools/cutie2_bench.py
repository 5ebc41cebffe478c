"""Throughput of BASELINE.json configs 4 and 5 (parity-test configurations, not the bench line):
  config 4  CUTIE2-ASPP, grid 128x64, V=28790, C=9, inference, batch swept
  config 5  CUTIE2-dilate, grid 256x128, forward and forward+backward(+Adam), B in {1, 4, 8}
Synthetic inputs as SURVEY.md 8(d): image ~U(-0.5, 0.5) at the grid's size, ps_1d ~U{0..h*W-2} with
token-less cells at 0.  Prints one JSON line per case."""
import argparse
import json
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from cutie_amd.runtime import Engine      # noqa: E402
from cutie_amd.train import Trainer       # noqa: E402


def run(variant, B, H, W, V, C, dtype, mode, steps, warmup):
    if variant == "aspp":
        from cutie_amd.model_cutie2_aspp import CUTIE2
    else:
        from cutie_amd.model_cutie2_dilate import CUTIE2
    p = types.SimpleNamespace(embedding_size=128, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIE2(V, C, p)
    net.build_loss()
    eng = Engine(net, "cuda:0", dtype=dtype, seed=0)
    g = torch.Generator().manual_seed(1)
    occ = torch.rand((B, H, W), generator=g) < 0.12
    grid = torch.where(occ, torch.randint(1, V - 1, (B, H, W), generator=g), torch.zeros((), dtype=torch.int64)).int().cuda()
    labels = torch.where(occ & (torch.rand((B, H, W), generator=g) < 0.25), torch.randint(1, C, (B, H, W), generator=g),
                         torch.zeros((), dtype=torch.int64)).int().cuda()
    image = (torch.rand((B, H, W, 3), generator=g) - 0.5).cuda()
    ps = torch.where(occ.reshape(B, -1), torch.randint(0, H * W - 1, (B, H * W), generator=g),
                     torch.zeros((), dtype=torch.int64)).int().cuda()
    tr = Trainer(eng, learning_rate=1e-4) if mode == "train" else None
    batch = {"grid_table": grid, "gt_classes": labels, "data_image": image, "ps_1d_indices": ps}

    def step():
        if mode == "train":
            tr.step(batch, fetch=False)
        else:
            eng.forward(grid, image=image, ps_1d_indices=ps)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"model": "CUTIE2-" + variant, "grid": [H, W], "batch": B, "dtype": dtype, "mode": mode,
                      "ms_per_step": round(1e3 * dt, 3), "docs_per_s": round(B / dt, 1),
                      "hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    del eng, tr
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--only", default="", help="'aspp32': config 4 at B = 32 alone, 'dilate8': config 5 training at B = 8 alone (for a profiler)")
    a = ap.parse_args()
    if a.only == "aspp32":
        run("aspp", 32, 128, 64, 28790, 9, a.dtype, "infer", a.steps, a.warmup)
        sys.exit(0)
    if a.only == "dilate8":
        run("dilate", 8, 256, 128, 28790, 9, a.dtype, "train", a.steps, a.warmup)
        sys.exit(0)
    for B in (32, 128, 512):
        run("aspp", B, 128, 64, 28790, 9, a.dtype, "infer", a.steps, a.warmup)
    for B in (1, 4, 8):
        run("dilate", B, 256, 128, 28790, 9, a.dtype, "infer", a.steps, a.warmup)
        run("dilate", B, 256, 128, 28790, 9, a.dtype, "train", a.steps, a.warmup)
