"""Data parallelism for the train step: one process per GPU, documents sharded by rank,
gradients averaged by a bucketed all-reduce (RCCL over xGMI on the GPU box; gloo in CPU tests).

The reference has no multi-GPU path (CUDA_VISIBLE_DEVICES="1", main_train_json.py:9); this is
new.  The gradient buffer is flat and ordered so that backward finishes it front to back
(runtime.Engine._layout_params), hence a bucket is a contiguous slice that can be reduced, on
the collective's own stream, while the earlier layers are still in backward.  xGMI is
point-to-point (7 links per GPU), so few, large buckets: 54.6 MB of f32 gradients go out as two
slices of the conv range (one per batched weight-gradient call of the backward pass) plus one slice for
biases + embedding, which also carries the embedding rows' share of the clip norm.
"""
import torch
import torch.distributed as dist


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_documents(docs, rank, world):
    """rank r takes documents r, r+world, ... of the global batch."""
    return list(docs[rank::world])


class GradBuckets(object):
    """Tracks which prefix of the flat gradient buffer is final and all-reduces it bucket-wise.

    ready(upto) - gradients in [0, upto) are final: launch every bucket that is now complete.
    finish()    - launch what is left (biases, embedding), wait for everything.
    The sum is left in the buffer; the optimizer applies 1/world (cutie_adam_step grad_scale)."""

    def __init__(self, flat, n_front, n_buckets=4, group=None, align=64, bounds=None):
        """bounds: explicit ends of the front-range slices (ascending, the last one n_front) - Trainer passes the points
        at which Engine.loss_backward announces a finished prefix, so that no slice waits for the announcement after
        its own; default: n_buckets equal slices."""
        self.flat, self.group = flat, group
        self._checked = False
        self.n_front = int(n_front)
        if bounds is not None:
            self.bounds = sorted(set(int(b) for b in bounds if 0 < int(b) < self.n_front)) + [self.n_front]
        else:
            step = max(align, (self.n_front // max(1, n_buckets) + align - 1) // align * align)
            self.bounds = list(range(step, self.n_front, step)) + [self.n_front]
        self.reset()

    def _adjoins(self, extra):
        f = self.flat
        return (extra.dtype == f.dtype and extra.device == f.device and extra.is_contiguous() and f.is_contiguous()
                and extra.untyped_storage().data_ptr() == f.untyped_storage().data_ptr()
                and extra.storage_offset() == f.storage_offset() + f.numel())

    def reset(self):
        self.next_lo, self.next_b, self.works = 0, 0, []

    def _launch(self, lo, hi):
        if hi > lo:
            if not self._checked:
                # device tensors go over RCCL only.  gloo stages them through the host on threads of its own; with two
                # ranks on one GPU that dead-locked intermittently in round 1 and the cause was never found, so the
                # path is gone rather than papered over: gloo is for the CPU tests (host tensors).
                if self.flat.is_cuda and dist.get_backend(self.group) != "nccl":
                    raise RuntimeError("GradBuckets: device gradients need the nccl (RCCL) backend, not %s"
                                       % dist.get_backend(self.group))
                self._checked = True
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, upto):
        while self.next_b < len(self.bounds) and self.bounds[self.next_b] <= upto:
            hi = self.bounds[self.next_b]
            self._launch(self.next_lo, hi)
            self.next_lo, self.next_b = hi, self.next_b + 1

    def finish(self, extra=None):
        """extra: a small tensor summed over the ranks along with the last slice (the embedding's share of the
        clip norm, see Trainer.step).  When it sits right behind `flat` in the same storage (Engine.grads_extra) the
        last slice simply reaches over it; otherwise it is a collective of its own."""
        hi = self.flat.numel()
        if extra is not None and self._adjoins(extra):
            store = self.flat.as_strided((hi + extra.numel(),), (1,))      # same storage, `extra` included
            if hi + extra.numel() > self.next_lo:
                self.works.append(dist.all_reduce(store[self.next_lo:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            extra = None
        else:
            self._launch(self.next_lo, hi)
        if extra is not None:
            self.works.append(dist.all_reduce(extra, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self.works:
            w.wait()
        self.reset()
