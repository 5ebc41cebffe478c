"""The train step of /root/reference/main_train_json.py:150-158,228-257 on the HIP engine.

Reference semantics kept: feed (grid_table int32 [B,H,W,1], gt_classes [B,H,W]) -> fetch
(model_loss, regularization_loss, total_loss, logits, softmax); tf.gradients of total_loss;
tf.clip_by_global_norm(grads, 10.0); tf.train.AdamOptimizer(lr) with TF's bias-correction form;
learning-rate decay x0.1 every lr_decay_step iterations (:225-226).  The two partial_run phases
collapse into one step because ghm_weights is all ones when use_ghm=0 (:249-251).

New: data parallelism (cutie_amd.parallel) - each rank runs the step on its shard of documents;
hard negative mining is per replica (SURVEY.md 8e caveat); the clip norm is TF's at every world size (the embedding
enters through its per-cell rows, summed over the ranks).  The losses a rank reports are its own shard's.
"""
import math
import os

import torch

from .parallel import GradBuckets, dist_info


class Trainer(object):
    def __init__(self, engine, learning_rate=1e-4, clip_norm=10.0, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 lr_decay_step=13000, lr_decay_factor=0.1, n_buckets=2, check_inputs=False,
                 scale_growth_interval=1000, gc_freeze=True):
        self.eng = engine
        self.lr = float(learning_rate)
        self.clip, self.b1, self.b2, self.eps = float(clip_norm), beta1, beta2, epsilon
        self.lr_decay_step, self.lr_decay_factor = lr_decay_step, lr_decay_factor
        self.iter = 0          # the reference's `iter` (drives lr decay)
        self.t = 0             # Adam's step count
        self.check_inputs = check_inputs
        # After the first step everything long-lived exists (torch, the engine's plans, the descriptor tables): move it out
        # of the garbage collector's reach (gc.freeze).  A full collection over that object graph takes ~39 ms of HOST time
        # on the MI355X box (tools/step_trace.py) - five device steps; it is absorbed while the host runs ahead of the
        # device queue and stalls the device when it does not (the "38 ms start-up step" of rounds 1 and 2: a gen-2 pass
        # in a step that the caller synchronised).  Frozen objects are never scanned again; new garbage is collected as before.
        self.gc_freeze = bool(gc_freeze)
        self._frozen = False
        self.scale_growth_interval = int(scale_growth_interval)
        self.m = torch.zeros_like(engine.params_flat)
        self.v = torch.zeros_like(engine.params_flat)
        self.rank, self.world = dist_info()
        # CUTIE_DP_FORCE_BUCKETS=1: take the bucketed all-reduce path in a one-rank group too (rehearsal
        # of the multi-GPU step on a single GPU; a one-rank all-reduce leaves the gradients as they are)
        import torch.distributed as dist
        forced = os.environ.get("CUTIE_DP_FORCE_BUCKETS") == "1" and dist.is_available() and dist.is_initialized()
        # the conv range goes out in the slices that backward finishes it in (Engine.wgrad_flush_points: two batched
        # weight-gradient calls); n_buckets only matters for an engine without such points
        pts = engine.wgrad_flush_points() if hasattr(engine, "wgrad_flush_points") and getattr(engine, "wgrad_batch", False) else None
        self.buckets = (GradBuckets(engine.grads_flat, engine.n_conv_w, n_buckets, bounds=pts)
                        if (self.world > 1 or forced) else None)

    def step(self, batch, fetch=True):
        """batch: {'grid_table', 'gt_classes'[, 'data_image', 'ps_1d_indices']} as next_batch() returns
        (data_loader_json.py:204-207).  Returns the reference's fetches as a dict."""
        eng = self.eng
        if self.iter != 0 and self.lr_decay_step and self.iter % self.lr_decay_step == 0:
            self.lr *= self.lr_decay_factor
        P = eng.forward(batch["grid_table"], image=batch.get("data_image"), ps_1d_indices=batch.get("ps_1d_indices"),
                        labels=batch["gt_classes"])
        if self.check_inputs:
            eng.check_inputs(P)            # bad ids / labels / sampler indices, before anything is trained on them
        emb_sq = None
        if self.buckets is not None:
            # each bucket's all-reduce starts as soon as backward has finished its slice and runs
            # on the collective's stream under the remaining layers' dgrad/wgrad kernels
            eng.loss_backward(P, on_ready=self.buckets.ready, defer_l2=True)
            # tf.clip_by_global_norm takes the embedding gradient as IndexedSlices: one row per grid cell, NOT summed
            # per id.  For the global batch those rows are the concatenation of every rank's rows, so
            # their squared norm is the SUM of the ranks' local ones: one more float in the last collective, and the
            # clip is the same function of the global batch at every world size.
            emb_sq = eng.embedding_rows_sumsq(P)
            self.buckets.finish(extra=emb_sq)
        else:
            eng.loss_backward(P, defer_l2=True)
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        # L2 term (world * wd * w beside the summed gradient), clip, Adam and the repack of the conv operands in two passes
        # over the flat buffers.  The host's t runs ahead of m and v by the steps the device skipped - an f16 overflow, or
        # conv tiles that gave up waiting for each other (P.late): the kernel takes them out of the bias correction
        # (opt_state[3]) and leaves everything untouched on such a step; no host read-back
        gn = eng.apply_update(P, self.m, self.v, lr_t, self.b1, self.b2, self.eps, self.clip, world=self.world, t=self.t,
                              emb_rows_sq=emb_sq, scale_growth_interval=self.scale_growth_interval)
        if self.check_inputs:
            eng.check_inputs(P)            # late tiles of the whole step, backward included (the device skipped it already)
        self.iter += 1
        if self.gc_freeze and not self._frozen:
            import gc
            gc.collect()
            gc.freeze()
            self._frozen = True
        if not fetch:
            return None
        ml, rg, tl = eng.losses(P)
        return {"model_loss": ml, "regularization_loss": rg, "total_loss": tl,
                "logits": eng.get(P, "cls_logits"), "softmax": P.probs, "global_norm": math.sqrt(float(gn.item())) / self.world}      # of the mean gradient: what the clip sees
