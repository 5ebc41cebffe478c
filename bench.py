#!/usr/bin/env python3
"""Headline benchmark: docs/sec of the CUTIE-ASPP hot path, 64x64 grid, fwd+bwd (BASELINE.json).

One step = one pass of the hot path over one batch of synthetic documents that are already
resident in HBM as packed OCR tokens: grid build (cutie_grid_build) -> embedding gather -> the
dense/atrous conv stack with instance norm and the ASPP concat -> softmax-CE + hard negative
mining -> backward (dgrad, wgrad, IN backward, embedding scatter-add) -> L2, global-norm clip,
Adam, weight repack.  N > 1: one process per GPU, documents sharded by rank (weak scaling,
32 documents per GPU), gradients all-reduced over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (the MFMA conv forward / data-gradient kernel, timed live with HIP events on
the launch stream) and `cpu_baseline` (the oracle's PyTorch-CPU restatement on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BF16_PEAK_TFLOPS = 2516.6     # MI355X dense bf16 MFMA: 256 CU x 2.4 GHz x 4096 FLOP/clk/CU
PRIME = 8                     # untimed steps before the warm-up: the device's first ~6 steps run 9.4 -> 7.9 ms (clock ramp, cold caches)
GF_PER_DOC_INBOUNDS = 231.743  # CUTIE-ASPP 64x64 fwd+bwd, in-bounds taps only


def conv_launch_flops(segs, B, H, W, n_out):
    """In-bounds multiply-add FLOPs of one conv_fwd launch (zero-padding taps excluded)."""
    total = 0
    for s in segs:
        cin = s.get("cin_real", s["cin"])
        hsum = sum(max(0, H - abs((i - s["kh"] // 2) * s["dil"])) for i in range(s["kh"]))
        wsum = sum(max(0, W - abs((j - s["kw"] // 2) * s["dil"])) for j in range(s["kw"]))
        total += 2 * B * hsum * wsum * cin * n_out
    return total


class ConvTimer(object):
    """Wraps ops.conv_fwd, ops.conv_in_fwd and ops.conv_dgrad_in_bwd: brackets every forward / data-gradient conv launch
    with HIP events on the launch stream.  A launch that also carries the instance norm (forward) or its gradient (the
    one-launch forms) is timed as a whole and credited with the conv's FLOPs only; the layers whose call falls back to
    the two-call form (dilation 16) have the norm pass inside their bracket - the fraction reads low, never high."""

    def __init__(self, ops):
        self.ops, self.orig, self.orig_bwd, self.orig_in_fwd = ops, ops.conv_fwd, ops.conv_dgrad_in_bwd, ops.conv_in_fwd
        self.events, self.flops, self.kinds, self.labels, self.enabled = [], [], [], [], False
        ops.conv_fwd = self
        ops.conv_dgrad_in_bwd = self.dgrad_in_bwd
        ops.conv_in_fwd = self.in_fwd

    def in_fwd(self, segs, a, y, n_out, *args, **kw):
        if not self.enabled:
            return self.orig_in_fwd(segs, a, y, n_out, *args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self.orig_in_fwd(segs, a, y, n_out, *args, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.flops.append(conv_launch_flops(segs, y.shape[0], y.shape[1], y.shape[2], n_out))
        self.kinds.append("forward + instance norm" if r else "forward, then in_apply")
        self.labels.append(self._label(segs, n_out))
        return r

    def dgrad_in_bwd(self, segs, dy, n_out, *a, **kw):
        if not self.enabled:
            return self.orig_bwd(segs, dy, n_out, *a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self.orig_bwd(segs, dy, n_out, *a, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.flops.append(conv_launch_flops(segs, dy.shape[0], dy.shape[1], dy.shape[2], n_out))
        self.kinds.append("data gradient + norm gradient")
        self.labels.append(self._label(segs, n_out))
        return r

    # Timing events serialise the queue (each one is a barrier packet): bracketing all 28 conv
    # launches of every step costs ~6 % of the step.  So only every EVERY-th timed step is
    # instrumented; `value` stays within ~1 % of an uninstrumented run.
    EVERY = 5

    def __call__(self, segs, out, n_out, **kw):
        if not self.enabled:
            return self.orig(segs, out, n_out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self.orig(segs, out, n_out, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.flops.append(conv_launch_flops(segs, out.shape[0], out.shape[1], out.shape[2], n_out))
        self.kinds.append("conv alone")
        self.labels.append(self._label(segs, n_out))
        return r

    @staticmethod
    def _label(segs, n_out):
        return "+".join("%s%dx%d d%d %d" % ("T" if s.get("flip") else "", s["kh"], s["kw"], s["dil"], s["cin"]) for s in segs) + " -> %d" % n_out

    def launch_table(self, peak_tflops):
        """--launch-table: every conv launch of a step by its position in the step (mean over the instrumented steps)"""
        if not self.events:
            return []
        first = self.labels[0]
        per = next((i for i in range(1, len(self.labels)) if self.labels[i] == first and self.kinds[i] == self.kinds[0]
                    and len(self.labels) % i == 0 and self.labels[:i] == self.labels[i:2 * i]), len(self.labels))
        rows = []
        for j in range(per):
            ms = [a.elapsed_time(b) for (a, b) in self.events[j::per]]
            m = sum(ms) / len(ms)
            rows.append("%2d %-30s %-34s %7.1f us  %.3f" % (j, self.kinds[j], self.labels[j], 1e3 * m,
                                                         self.flops[j] / (m * 1e-3) / 1e12 / peak_tflops))
        return rows

    def by_kind(self, peak_tflops):
        """the same quotient per kind of bracket: only 'conv alone' is a conv kernel by itself (round 1's figure was all of
        that kind); the others carry a norm pass or its gradient inside the bracket, which earns no FLOPs"""
        out = {}
        for (a, b), fl, kind in zip(self.events, self.flops, self.kinds):
            e = out.setdefault(kind, [0.0, 0.0, 0])
            e[0] += a.elapsed_time(b)
            e[1] += fl
            e[2] += 1
        return {k: {"launches": n, "avg_launch_ms": round(ms / n, 4), "frac": round(fl / (ms * 1e-3) / 1e12 / peak_tflops, 4)}
                for k, (ms, fl, n) in out.items() if ms > 0}

    def summary(self):
        ms = [a.elapsed_time(b) for a, b in self.events]
        return sum(ms), sum(self.flops), len(ms)


def sample_clock(ops, dev, step, fence, steps=8):
    """The shader clock the chip HOLDS inside the dominant kernel, measured in the kernel: workgroup 0 of every wide conv
    launch of `steps` extra, untimed steps adds its K loop's shader-clock cycles (s_memtime) and 100 MHz ticks
    (s_memrealtime) to a probe buffer (cutie_conv_clock_probe)."""
    probe = torch.zeros(3, dtype=torch.int64, device=dev)
    for _ in range(6):      # the device may have idled behind the timed region (CPU baseline): back to the steady state first
        step()
    fence()
    ops.conv_clock_probe(probe)
    try:
        for _ in range(steps):
            step()
        fence()
    finally:
        ops.conv_clock_probe(None)
    cycles, ticks, launches = (int(v) for v in probe.cpu())
    if ticks <= 0:
        return {"mhz": None, "note": "no wide conv launch reported"}
    return {"mhz": round(100.0 * cycles / ticks, 1), "mhz_nominal": 2400, "launches": launches,
            "note": "shader-clock cycles / 100 MHz ticks over the K loop of workgroup 0 of every conv_wide launch of %d untimed "
                    "steps behind the timed region; the 2516.6 TFLOP/s peak assumes 2400 MHz" % steps}


def kernel_source_sha():
    """sha256 over the kernel sources: a PMC profile only describes the kernels it was taken with"""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv_wide.hip", "conv_common.h", "conv_mfma.hip", "common.h", "embed.hip", "loss.hip"):
        with open(os.path.join(REPO, "cutie_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_profile():
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc.json")))
    if not files:
        return None, None, "no profiles/*_pmc.json"
    with open(files[-1]) as f:
        prof = json.load(f)
    rel = os.path.relpath(files[-1], REPO)
    if prof.get("source_sha") != kernel_source_sha():
        return None, rel, "%s was taken with other kernel sources (source_sha %s != %s): not reported" % (
            rel, prof.get("source_sha"), kernel_source_sha())
    return prof, rel, None


class CallTimer(object):
    """HIP events around the HBM-bound calls of the path north_star names - the embedding gather and its scatter-add, the
    softmax-CE head forward and backward - on the instrumented steps, in place (inside real steps: their operands are as
    cold as they are in training, not re-read from a warm cache by a micro-benchmark loop)."""
    # call -> (kernels of the call as the profile names them, algorithmic bytes per cell by SURVEY.md 8d for 16-bit activations)
    CALLS = {
        "embed_fwd": (("embed_fwd_kernel",), lambda E, C: 4 + 2 * E * 2,
                      "gather: id + table row + stored row (SURVEY.md 8d: 4 + 2*E*s)"),
        "embed_bwd": (("embed_bwd_kernel", "embed_bwd_fold_kernel"), lambda E, C: 4 + E * 2 + E * 4,
                      "scatter-add: id + gradient row read + f32 row added (SURVEY.md 8d: 4 + E*s + E*4; rows of one id are "
                      "summed per workgroup first, so far fewer bytes reach the table)"),
        "head_fwd": (("head_conv_kernel", "head_norm_ce_kernel"), lambda E, C: 64 * 2 + 3 * C * 4 + 4 + 4 + C * 4,
                     "logits layer (64 x C MFMA) + instance norm + softmax + CE: x row read, a written and read, logits, probs, "
                     "label, ce (SURVEY.md 8d counts C*4 read + 4 + 4 + C*4 probs for the loss alone)"),
        "head_bwd": (("hnm_hist_kernel", "hnm_tie_kernel", "hnm_weight_kernel", "head_bwd_kernel"),
                     lambda E, C: 3 * (4 + 1) + (4 + 1 + 4) + 3 * C * 4 + 4 + 64 * 2,
                     "hard-negative select (three passes over ce + labels), weights, d(loss)/d(logits) and the logits layer's "
                     "norm + ReLU gradient: a, logits, probs read, dz row written"),
    }

    def __init__(self, ops):
        self.ops, self.enabled = ops, False
        self.ms = {k: [] for k in self.CALLS}
        self.events = {k: [] for k in self.CALLS}
        for name in self.CALLS:
            setattr(ops, name, self._wrap(name, getattr(ops, name)))

    def _wrap(self, name, fn):
        def call(*a, **kw):
            if not self.enabled:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            self.events[name].append((e0, e1))
            return r
        return call

    def report(self, n_cells, E, C):
        prof, rel, why = _pmc_profile()
        out = []
        for name, (kernels, per_cell, what) in self.CALLS.items():
            ev = self.events[name]
            if not ev:
                continue
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            alg = per_cell(E, C) * n_cells
            e = {"call": name, "what": what, "kernels": list(kernels), "ms": round(ms, 4), "bytes_algorithmic": int(alg),
                 "GBps_algorithmic": round(alg / (ms * 1e-3) / 1e9, 1)}
            pmc = None
            if prof is not None:
                pmc = 0
                for kn in kernels:
                    hit = [v for k, v in prof["kernels"].items() if kn in k]
                    if not hit:
                        pmc = None
                        break
                    # gfx950 tallies a 16-byte-per-lane read stream at half its bytes (MI355X_MICROARCH.md, HBM): the kernels
                    # that read float4 / 8 x f16 per lane take the doubled FETCH_SIZE, the 4- and 8-byte readers the raw one
                    wide = kn.startswith(("embed_fwd", "head_"))
                    pmc += sum(h["hbm_bytes_per_launch" if wide else "hbm_bytes_raw"] for h in hit)
            e["bytes_pmc"] = pmc
            if pmc is not None:
                e["GBps"] = round(pmc / (ms * 1e-3) / 1e9, 1)
                e["frac_of_8TBps"] = round(pmc / (ms * 1e-3) / 8e12, 4)
            else:
                e["GBps"], e["frac_of_8TBps"] = None, None
                e["bytes_pmc_source"] = why
            out.append(e)
        return out, rel


def pmc_traffic(dtype):
    """HBM bytes per conv_wide launch from the newest committed PMC profile (PMC passes cannot run inside the timed
    bench).  Reported only when that profile was taken with the kernel sources of this tree (its `source_sha`);
    otherwise null, with the reason."""
    prof, rel, why = _pmc_profile()
    if prof is None:
        return None, why
    # one kernel instance per epilogue mode (plain / norm sums / backward rendezvous / forward rendezvous): launch-weighted mean
    code = {"bf16": 1, "f16": 2}.get(dtype, -1)
    ks = [v for name, v in prof["kernels"].items() if name.startswith("conv_wide_kernel<%d, 256," % code)]
    if not ks:
        return None, "%s holds no conv_wide_kernel<%s, 256, *>" % (rel, dtype)
    n = sum(k["launches"] for k in ks)
    per_launch = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in ks) / n
    return per_launch, ("%s: (2*FETCH_SIZE + WRITE_SIZE)*1024 per conv_wide_kernel<%s,256,*> launch, mean over its four epilogue "
                        "modes weighted by launches (tools/gpu_pmc.sh); algorithmic bytes 136 MB forward, 203 MB data gradient "
                        "with the fused norm gradient" % (rel, dtype))


def cpu_baseline(num_classes, V, E, H, W, packed_docs):
    """The oracle (PyTorch CPU, f32, all host threads) on a bounded sample of the same workload."""
    from oracle import cutie_oracle as co
    from oracle import grid_oracle as go
    from cutie_amd.synthetic import as_token_dicts
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    cores = torch.get_num_threads()
    docs = as_token_dicts(packed_docs)
    params = co.init_params(V, E, num_classes, seed=0, dtype=torch.float32)
    t0 = time.perf_counter()
    g = go.positional_mapping(docs, H, W, 1, num_classes)
    t_grid = time.perf_counter() - t0
    grid = torch.from_numpy(g["grid"]).long()
    labels = torch.from_numpy(g["labels"].astype(np.int64)).clamp(0, num_classes - 1)
    co.loss_and_grads(params, grid[:1], labels[:1])              # warm-up (1 doc)
    # bounded sample: 4-document passes (autograd keeps ~0.4 GB per document) until ~12 s are spent
    n_docs, t_model, k = 0, 0.0, 0
    while t_model < 12.0 and n_docs < 64:
        lo = (4 * k) % len(docs)
        t0 = time.perf_counter()
        co.loss_and_grads(params, grid[lo:lo + 4], labels[lo:lo + 4])
        t_model += time.perf_counter() - t0
        n_docs += int(grid[lo:lo + 4].shape[0])
        k += 1
    per_doc = t_grid / len(docs) + t_model / n_docs
    return {"value": round(1.0 / per_doc, 4), "unit": "docs/s", "cores": cores, "kind": "port",
            "sample": "oracle restatement (PyTorch CPU f32 autograd + python grid builder): %d documents fwd+bwd in "
                      "4-document passes (%.1f s) after a 1-document warm-up, grid build on %d; TensorFlow 1.12 itself "
                      "cannot run here" % (n_docs, t_model, len(docs)),
            "grid_build_ms_per_doc": round(1e3 * t_grid / len(docs), 3), "model_s_per_doc": round(t_model / n_docs, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="documents per GPU")
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"],
                    help="f16 is the headline path: the 16-bit path that meets the 2e-2 logit tolerance at the bf16 MFMA rate")
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--cols", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-clock", action="store_true", help="skip the shader-clock samples behind the timed region")
    ap.add_argument("--launch-table", action="store_true", help="stderr: every conv launch of a step with its event time")
    ap.add_argument("--no-alt", "--no-f16", dest="no_alt", action="store_true",
                    help="skip the second pass on the other 16-bit element type (bf16 when the headline is f16)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    # under torch.distributed.run (RANK set) the RCCL group is always formed, also for one rank, so
    # that the 1-GPU box can rehearse the exact code path the 2/4/8-GPU launches take
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from cutie_amd import ops
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.synthetic import make_documents
    from cutie_amd.train import Trainer

    V, E, C, H, W, B = 20000, 128, 18, args.rows, args.cols, args.batch
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIERes(V, C, p)
    net.build_loss()
    eng = Engine(net, dev, dtype=args.dtype, seed=0)          # same seed on every rank: identical replicas
    trainer = Trainer(eng, learning_rate=1e-4)
    packed = make_documents(B, C, seed=1234 + 1000 * rank)
    tok = {k: torch.from_numpy(v).to(dev) for k, v in packed.items()}
    timer = ConvTimer(ops)
    calls = CallTimer(ops)

    def step():
        g = ops.grid_build(tok["tok_box"], tok["tok_id"], tok["tok_ecid"], tok["doc_extent"], tok["doc_off"], H, W, 1)
        trainer.step({"grid_table": g["grid"], "gt_classes": g["labels"]}, fetch=False)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Start-up: the device's own step time is flat from about the sixth step on (tools/step_trace.py: 9.40 8.44 8.21 8.04
    # 7.88 7.81 7.79 7.73 7.72 ... ms).  The "38 ms step" of rounds 1 and 2 was a generation-2 pass of Python's garbage
    # collector (39.5 ms of host time over torch's object graph): Trainer freezes the long-lived objects after its first
    # step, so no such pass falls into the timed region.  PRIME untimed steps come on top of the W warm-up steps.
    for _ in range(PRIME):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    overlap = eng.overlap_wgrad
    for k in range(args.steps):
        timer.enabled = (not args.no_kernel_timing) and k % ConvTimer.EVERY == 0
        calls.enabled = timer.enabled
        # an instrumented step keeps the weight gradients on the launch stream: the events must see one
        # kernel at a time (beside a weight gradient a conv launch shares the chip and reads longer)
        eng.overlap_wgrad = overlap and not timer.enabled
        step()
    fence()
    dt = time.perf_counter() - t0
    timer.enabled = False
    calls.enabled = False
    eng.overlap_wgrad = overlap
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    plan = eng._plans[(B, H, W, 0)]
    # a number taken over steps the device declared invalid (a conv tile gave up waiting for its sample's other tiles and
    # the optimizer skipped the step) or over bad inputs is no measurement: raise instead of printing a line
    eng.check_inputs(plan)
    if eng.steps_invalid():
        raise RuntimeError("%d of the steps were skipped as invalid on the device (late conv tiles): no bench line" % eng.steps_invalid())
    loss = eng.losses(plan)

    if rank == 0:
        docs_per_s = world * B * args.steps / dt
        out = {"metric": "docs/sec CUTIE-ASPP 64x64 fwd+bwd", "value": round(docs_per_s, 2), "unit": "docs/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic",
               "config": {"workload": "CUTIE-ASPP fwd+bwd+Adam incl. grid build, %dx%d grid, emb %d, vocab %d, %d classes, "
                                      "%d synthetic docs per GPU (BASELINE.json configs[1])" % (H, W, E, V, C, B),
                          "global_batch": world * B, "parallelism": "dp%d" % world, "prime_steps": PRIME,
                          "whole_step_mfma_frac": round(docs_per_s / world * GF_PER_DOC_INBOUNDS * 1e9 / (BF16_PEAK_TFLOPS * 1e12), 4),
                          "final_total_loss": round(loss[2], 5)}}
        out["config"]["loss_scale"], out["config"]["steps_skipped"] = eng.loss_scale_value()
        out["config"]["steps_invalid"] = eng.steps_invalid()
        if not args.no_kernel_timing:
            ms, fl, n = timer.summary()
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            peak = BF16_PEAK_TFLOPS if args.dtype in ("bf16", "f16") else 157.3    # f16 and bf16 MFMA run at one rate
            traffic, traffic_src = pmc_traffic(args.dtype)
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "conv_wide_kernel + conv_kernel (forward + data-gradient implicit GEMM, instance norm / its gradient inside the launch)", "launches": n,
                               "avg_launch_ms": round(ms / max(n, 1), 4), "by_kind": timer.by_kind(peak),
                               "timed_steps": "every %d-th of the %d timed steps%s" % (
                                   ConvTimer.EVERY, args.steps,
                                   ", run on one stream (the others run the weight gradients on a second stream)" if overlap else ""),
                               "share_of_step": round(ms / max(1, -(-args.steps // ConvTimer.EVERY)) / (1e3 * dt / args.steps), 4)}
            # the HBM-bound kernels north_star names (gather, loss): event time inside real steps, bytes from the PMC profile
            out["roofline_hbm"], _ = calls.report(B * H * W, E, C)
            if args.launch_table:
                print("\n".join(timer.launch_table(peak)), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(C, V, E, H, W, packed)
    if not args.no_clock:
        # the peak is the vendor's 2.4 GHz figure - print the shader clock the chip actually HOLDS under this
        # step, measured inside the conv kernels themselves on extra, untimed steps behind the timed region (amdsmi's "current
        # clock" read 1147, 1613 and 2250 MHz for the same step on three boxes: a DPM level, not the clock).  EVERY rank runs
        # the extra steps (under data parallelism a step holds collectives); rank 0 reports its own GPU's figure.
        clk = sample_clock(ops, dev, step, fence)
        if rank == 0:
            out["config"]["sclk_under_load"] = clk
            if clk.get("mhz"):
                # the same whole-step fraction against the MFMA rate at the clock the chip holds (peak x held / 2400 MHz)
                out["config"]["whole_step_mfma_frac_at_held_clock"] = round(
                    out["config"]["whole_step_mfma_frac"] * clk["mhz_nominal"] / clk["mhz"], 4)
    if args.dtype in ("f16", "bf16") and not args.no_alt:
        # the same step on the other 16-bit element type (same tiles, same MFMA rate): bf16 beside the f16
        # headline.  bf16's 8 significant bits put it outside the 2e-2 logit tolerance (tests/test_model_gpu.py),
        # so it is reported beside the headline number, never instead of it.
        alt = "bf16" if args.dtype == "f16" else "f16"
        del trainer, eng
        eng = Engine(net, dev, dtype=alt, seed=0)
        trainer = Trainer(eng, learning_rate=1e-4)
        for _ in range(PRIME + args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
        fence()
        dt16 = time.perf_counter() - t0
        eng.check_inputs(eng._plans[(B, H, W, 0)])
        if eng.steps_invalid():
            raise RuntimeError("%s pass: %d steps skipped as invalid on the device" % (alt, eng.steps_invalid()))
        if use_dist:
            t = torch.tensor([dt16], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt16 = float(t.item())
        if rank == 0:
            S, skipped = eng.loss_scale_value()
            out[alt + "_path"] = {"value": round(world * B * args.steps / dt16, 2), "unit": "docs/s",
                                  "ms_per_step": round(1e3 * dt16 / args.steps, 3), "loss_scale": S, "steps_skipped": skipped,
                                  "note": "same workload on the %s element type; no kernel timing events" % alt}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
