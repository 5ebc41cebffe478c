"""Where does a slow step come from?  Per step, without synchronising in between: host time to enqueue it, device time
between its first and last kernel (events), python gc passes that fell into it, allocator growth.
    python tools/step_trace.py [steps] [gc|nogc|freeze]"""
import gc
import sys
import time
import types

import torch

sys.path.insert(0, ".")
from cutie_amd import ops
from cutie_amd.model_cutie_aspp import CUTIERes
from cutie_amd.runtime import Engine
from cutie_amd.synthetic import make_documents
from cutie_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
mode = sys.argv[2] if len(sys.argv) > 2 else "gc"
V, E, C, H, W, B = 20000, 128, 18, 64, 64, 32
p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
net = CUTIERes(V, C, p)
net.build_loss()
eng = Engine(net, "cuda:0", dtype="f16", seed=0)
tr = Trainer(eng, learning_rate=1e-4)
packed = make_documents(B, C, seed=1234)
tok = {k: torch.from_numpy(v).cuda() for k, v in packed.items()}

gc_log = []
t_gc = [0.0]


def on_gc(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        gc_log.append((cur[0], info["generation"], 1e3 * (time.perf_counter() - t_gc[0]), info["collected"]))


cur = [0]
gc.callbacks.append(on_gc)
if mode == "nogc":
    gc.disable()


def step():
    g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], H, W, 1)
    tr.step({"grid_table": g["grid"], "gt_classes": g["labels"]}, fetch=False)


step()
torch.cuda.synchronize()
if mode == "freeze":
    gc.collect()
    gc.freeze()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host, reserved = [], []
t_start = time.perf_counter()
ev[0].record()
for k in range(steps):
    cur[0] = k
    t0 = time.perf_counter()
    step()
    ev[k + 1].record()
    host.append(1e3 * (time.perf_counter() - t0))
    reserved.append(torch.cuda.memory_reserved())
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t_start)
dev = [ev[k].elapsed_time(ev[k + 1]) for k in range(steps)]
print("mode %s: %d steps, wall %.1f ms = %.3f ms/step; device median %.3f max %.3f; host enqueue median %.3f max %.3f"
      % (mode, steps, wall, wall / steps, sorted(dev)[steps // 2], max(dev), sorted(host)[steps // 2], max(host)))
med = sorted(dev)[steps // 2]
for k in range(steps):
    if dev[k] > 1.5 * med or host[k] > 3 * sorted(host)[steps // 2]:
        print("  step %3d: device %.2f ms, host enqueue %.2f ms, reserved %.0f MB%s" % (
            k, dev[k], host[k], reserved[k] / 2 ** 20,
            "  (allocator grew)" if k and reserved[k] != reserved[k - 1] else ""))
print("gc passes: " + ", ".join("step %d gen%d %.1f ms (%d collected)" % e for e in gc_log if e[1] >= 1 or e[2] > 1.0))
print("device ms: " + " ".join("%.2f" % t for t in dev))
