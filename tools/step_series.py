"""Per-step wall time of the bench step over the first N steps (looks for start-up transients)."""
import sys, time, types
import torch
sys.path.insert(0, ".")
from cutie_amd import ops
from cutie_amd.model_cutie_aspp import CUTIERes
from cutie_amd.runtime import Engine
from cutie_amd.synthetic import make_documents
from cutie_amd.train import Trainer

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
V, E, C, H, W, B = 20000, 128, 18, 64, 64, 32
p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
net = CUTIERes(V, C, p); net.build_loss()
eng = Engine(net, "cuda:0", dtype="f16", seed=0)
tr = Trainer(eng, learning_rate=1e-4 if mode == "train" else 0.0)
packed = make_documents(B, C, seed=1234)
tok = {k: torch.from_numpy(v).cuda() for k, v in packed.items()}
ts = []
for k in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], H, W, 1)
    tr.step({"grid_table": g["grid"], "gt_classes": g["labels"]}, fetch=False)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(mode, " ".join("%.2f" % t for t in ts))
