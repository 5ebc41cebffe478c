"""Engine: compiles a cutie_amd.model_framework graph into HIP kernel launches.

What the reference delegates to a tf.Session (/root/reference/main_train_json.py:185-186,
:231-257; main_evaluate_json.py:99) happens here: variable creation/initialisation, buffer
planning for a given batch/grid shape, the forward pass, the loss, and the backward pass.
All arithmetic is in libcutie_hip.so; PyTorch only owns the device memory.

Memory plan (sized for 288 GB of HBM: nothing is recomputed, nothing is re-laid-out):
  * parameters, gradients and Adam moments are three flat f32 buffers with one layout
    [conv kernels, reverse creation order | biases | embedding]; the L2 term, the global
    norm, Adam and the RCCL all-reduce each run over a contiguous range of them;
  * every concat is one wide NHWC buffer whose producers write their channel slice
    (model_framework.py:294 needs no kernel); channel counts are padded to 64 with zeros so
    that the K dimension of every implicit GEMM is whole;
  * per conv layer: `a` = relu(conv + b) before the instance norm (its sign is the ReLU
    mask), `y` = normalised output, f32 moments; backward adds `dy` and `dz`.
A tensor with several consumers gets its gradient from ONE multi-segment data-gradient
launch over all consuming convs (also through concats, by weight-row offset).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import capi, ops
from .capi import BF16, F16, F32, CutieError


def _pad(c, m=64):
    return (c + m - 1) // m * m


class _Plan(object):
    pass


class Engine(object):
    def __init__(self, network, device="cuda:0", dtype="f16", seed=0, loss_scale=None, overlap_wgrad=None, max_plans=4):
        """dtype: 'f16' (the headline path: IEEE half activations and MFMA operands, f32 accumulation, statistics
        and parameters; meets the 2e-2 logit tolerance; backward runs under a dynamic loss scale kept on the device,
        initial value `loss_scale`, default 4096), 'bf16' (same tiles and MFMA rate, 8 significant bits: outside the
        tolerance on deep stacks, kept as an option) or 'f32' (parity mode, 1e-4)."""
        capi.lib()   # fail loudly, before anything else, when the HIP library is absent
        if not torch.cuda.is_available():
            raise CutieError("cutie_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.net = network
        self.dev = torch.device(device)
        self.dt = capi.parse_dtype(dtype)
        self.tdt = capi.torch_dtype(self.dt)
        self.bke = 32 if self.dt == F32 else 64
        if loss_scale is None and self.dt == F16:
            loss_scale = 4096.0
        with torch.cuda.device(self.dev):
            # optimizer state on the device {S, 1/S, good steps, steps skipped, of them invalid, ...}: every element type has
            # one (the skipped-step count re-bases Adam's bias correction); only f16 scales its backward pass by S
            self.opt_state = ops.loss_scale_state(loss_scale if loss_scale else 1.0, self.dev)
            self.loss_scale = self.opt_state if loss_scale else None
        # Optional: backward on two HIP streams - the weight gradients (MFMA-bound, needed only by the optimizer
        # and the all-reduce) beside the data-gradient / norm-backward chain (whose norm passes are HBM-bound).
        # Off by default: measured on the MI355X (tools/gpu_overlap.sh, alternating runs on one box) it is 0.45 %
        # SLOWER than one stream (3914 / 3916 vs 3932 / 3933 docs/s) - conv_wide and wgrad5 each fill a CU's LDS,
        # so the two matrix kernels only take turns, and the norm passes gain nothing beside them.
        # CUTIE_OVERLAP_WGRAD=1 or overlap_wgrad=True turns it on.
        if overlap_wgrad is None:
            overlap_wgrad = os.environ.get("CUTIE_OVERLAP_WGRAD", "0") == "1"
        self.overlap_wgrad = bool(overlap_wgrad)
        self._side = None
        self.C = network.num_classes
        # buffer sets per (B, H, W, Hi), least recently used first.  The reference's loader draws a new grid shape for
        # almost every training batch (data_loader_json.py:660-694: max over the documents plus |N(0,16)| padding), and
        # one set is ~4 GB at B = 32, 64 x 64: only the `max_plans` most recently used shapes keep theirs.
        self._plans = OrderedDict()
        self.max_plans = max(1, int(max_plans))
        self._layout_params()
        self.initialize(seed)

    # ------------------------------------------------------------------ variables
    def _layout_params(self):
        specs = list(self.net.param_specs)
        conv_w = [s for s in specs if len(s.shape) == 4]
        biases = [s for s in specs if len(s.shape) == 1]
        embeds = [s for s in specs if len(s.shape) == 2]
        order = list(reversed(conv_w)) + list(reversed(biases)) + embeds
        off, self.offsets = 0, OrderedDict()
        for s in order:
            n = int(np.prod(s.shape))
            self.offsets[s.name] = (off, n, s.shape)
            off = _pad(off + n, 64)
            if s is conv_w[0]:
                self.n_conv_w = off      # creation-first conv is last in the reversed range
        self.n_dense = self.offsets[embeds[0].name][0] if embeds else off
        self.n_total = off
        self.conv_w_names = [s.name for s in conv_w]
        z = lambda: torch.zeros(self.n_total, dtype=torch.float32, device=self.dev)
        self.params_flat = z()
        # the gradient buffer carries 64 spare floats behind it: slot 0 holds this replica's share of the clip norm that is
        # not in the buffer itself (the embedding's per-cell rows, embedding_rows_sumsq), so that the LAST all-reduce slice
        # of the data-parallel step sums it over the ranks with the gradients - no collective of its own
        self._grads_store = torch.zeros(self.n_total + 64, dtype=torch.float32, device=self.dev)
        self.grads_flat = self._grads_store[:self.n_total]
        self.grads_extra = self._grads_store[self.n_total:self.n_total + 1]
        self.params = OrderedDict((s.name, self._view(self.params_flat, s.name)) for s in specs)
        self.grads = OrderedDict((s.name, self._view(self.grads_flat, s.name)) for s in specs)
        self.spec_by_name = OrderedDict((s.name, s) for s in specs)
        # packed conv operands (rebuilt by repack() whenever the parameters change)
        self.w_fwd, self.w_dg = {}, {}
        for s in conv_w:
            kh, kw, cin, cout = s.shape
            self.w_fwd[s.name] = torch.zeros((kh * kw, cout, _pad(cin, self.bke)), dtype=self.tdt, device=self.dev)
            self.w_dg[s.name] = torch.zeros((kh * kw, cin, _pad(cout, self.bke)), dtype=self.tdt, device=self.dev)

    def _view(self, flat, name):
        off, n, shape = self.offsets[name]
        return flat[off:off + n].view(shape)

    def initialize(self, seed=0):
        """tf.global_variables_initializer() analogue with the reference's initialisers
        (model_framework.py:46,178-179): embedding U(-1,1); kernels truncated normal(0, 0.01); biases 0."""
        g = torch.Generator().manual_seed(seed)
        for s in self.net.param_specs:
            if len(s.shape) == 2:
                p = torch.rand(s.shape, generator=g) * 2 - 1
            elif len(s.shape) == 4:
                p = torch.nn.init.trunc_normal_(torch.empty(s.shape), mean=0.0, std=0.01, a=-0.02, b=0.02, generator=g)
            else:
                p = torch.zeros(s.shape)
            self.params[s.name].copy_(p)
        self.repack()

    def load_params(self, named):
        """named: mapping '<layer>/weights' (HWIO) / '<layer>/biases' / 'embedding/weights' -> array."""
        for k, v in named.items():
            if k not in self.params:
                raise KeyError("unknown variable %s" % k)
            t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).to(torch.float32)
            if tuple(t.shape) != tuple(self.params[k].shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(t.shape), tuple(self.params[k].shape)))
            self.params[k].copy_(t)
        self.repack()

    def state_dict(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.params.items())

    def repack(self):
        """params -> MFMA operand layouts, all layers in one launch (descriptors built once)."""
        if getattr(self, "_pack_batches", None) is None:
            self._pack_batches = ops.pack_weights_batched(
                [(self.params[n], self.w_fwd[n], self.w_dg[n]) for n in self.conv_w_names], self.dt)
        else:
            ops.replay_pack(self._pack_batches, self.dt, self.dev)

    # ------------------------------------------------------------------ planning
    def _plan(self, B, H, W, Hi):
        key = (B, H, W, Hi)
        if key in self._plans:
            self._plans.move_to_end(key)
            return self._plans[key]
        while len(self._plans) >= self.max_plans:
            self._plans.popitem(last=False)      # its tensors go back to the allocator once the caller drops the plan too
        net, dev, tdt = self.net, self.dev, self.tdt
        P = _Plan()
        P.key, P.B, P.H, P.W, P.Hi = key, B, H, W, Hi
        nodes = net.nodes
        spatial = lambda n: (H, W) if n.domain == "grid" else (Hi, W)
        P.spatial = spatial
        consumers = {}
        for n in nodes:
            for i in n.inputs:
                consumers.setdefault(i.name, []).append(n)
        P.consumers = consumers

        # ---- placement of activation tensors: concat inputs live inside the concat buffer
        place, bufs = {}, {}
        P.copies = []        # (src name, dst view) for a tensor that feeds a second concat
        for n in nodes:
            if n.op != "concat":
                continue
            hs, ws = spatial(n)
            buf = torch.zeros((B, hs, ws, _pad(n.channels) + 64), dtype=tdt, device=dev)
            bufs[n.name] = buf
            off = 0
            for i in n.inputs:
                if i.name in place:
                    P.copies.append((i.name, n.name, buf[..., off:off + i.channels]))
                else:
                    place[i.name] = (n.name, off)
                off += i.channels
            place[n.name] = (n.name, 0)
        if Hi:
            bufs["data_image"] = torch.zeros((B, Hi, W, 64 + 64), dtype=tdt, device=dev)
            place["data_image"] = ("data_image", 0)
        for n in nodes:
            if n.op in ("conv", "embed", "global_pool", "positional_sampling") and n.name not in place:
                hs, ws = spatial(n)
                logits = n.name == net.loss_logits_name
                ch = _pad(n.channels, 4) if logits else _pad(n.channels) + 64
                bufs[n.name] = torch.zeros((B, hs, ws, ch), dtype=torch.float32 if logits else tdt, device=dev)
                place[n.name] = (n.name, 0)
        P.bufs, P.place = bufs, place

        def view(name, channels):
            b, off = place[name]
            return bufs[b][..., off:off + channels]
        P.view = view

        # ---- per conv layer state
        convs = [n for n in nodes if n.op == "conv"]
        P.convs = convs
        # instance-norm moments as per-tile partial sums [B][T][C][2] (slot stores, no float atomics)
        tiles = {n.name: ops.conv_stats_tiles(*spatial(n)) for n in convs}
        n_stats = sum(B * tiles[n.name] * n.channels * 2 for n in convs)
        P.stats_all = torch.zeros(n_stats, dtype=torch.float32, device=dev)
        P.a, P.stats, P.red, P.dz, P.dy, P.dbpart, P.sync, P.sync_fw, P.fused = {}, {}, {}, {}, {}, {}, {}, {}, {}
        so = 0
        for n in convs:
            hs, ws = spatial(n)
            P.a[n.name] = torch.zeros((B, hs, ws, _pad(n.channels)), dtype=tdt, device=dev)
            cnt = B * tiles[n.name] * n.channels * 2
            P.stats[n.name] = P.stats_all[so:so + cnt].view(B, tiles[n.name], n.channels, 2)
            so += cnt
        pooled = [n.inputs[0].channels for n in nodes if n.op == "global_pool"]
        P.pool_scratch = (torch.zeros(B * capi.POOL_TILES * max(pooled) * 2, dtype=torch.float32, device=dev)
                          if pooled else None)
        P.counters = torch.zeros(4, dtype=torch.int32, device=dev)     # one fill per forward clears all
        P.oob, P.bad, P.ps_oob = P.counters[0:1], P.counters[1:2], P.counters[2:3]
        # conv tiles that gave up waiting for their sample's other tiles in any launch of the step (forward or backward):
        # the optimizer skips the update on the device when it is non-zero (cutie_opt_step `invalid`)
        P.late = P.counters[3:4]
        N = B * H * W
        P.N = N
        # the logits head (1x1 conv + norm + softmax + CE + hard-negative select) as its own few launches: cutie_head_*
        P.head = self._head_node()
        if P.head is not None:
            Cp = _pad(self.C, 4)
            P.head_a = torch.zeros((B, H, W, Cp), dtype=torch.float32, device=dev)
            P.head_stats = torch.zeros((B, ops.head_tiles(H * W), self.C, 2), dtype=torch.float32, device=dev)
            P.head_red = torch.zeros((B, ops.head_chunks(H * W), self.C, 2), dtype=torch.float32, device=dev)
            P.head_ws = torch.empty(ops.head_workspace_bytes(B, H * W), dtype=torch.uint8, device=dev)
        P.ce = torch.zeros(N, dtype=torch.float32, device=dev)
        P.probs = torch.zeros((B, H, W, self.C), dtype=torch.float32, device=dev)
        P.weight = torch.zeros(N, dtype=torch.float32, device=dev)
        P.loss = torch.zeros(4, dtype=torch.float32, device=dev)
        P.zero_labels = torch.zeros(N, dtype=torch.int32, device=dev)
        P.grid_buf = torch.zeros((B, H, W), dtype=torch.int32, device=dev)
        P.labels_buf = torch.zeros((B, H, W), dtype=torch.int32, device=dev)
        P.grid, P.labels = P.grid_buf, P.labels_buf      # forward() points them at the caller's device tensors where it can
        P.ps = torch.zeros((B, H * W), dtype=torch.int32, device=dev) if Hi else None
        P.bwd_ready = False
        self._plans[key] = P
        return P

    # CUTIE_HEAD_FUSED=0: the logits layer and the loss on the general kernels (conv, in_stats, in_apply, softmax_ce, hnm, ...)
    head_fused = os.environ.get("CUTIE_HEAD_FUSED", "1") != "0"

    def _head_node(self):
        """the logits conv node when cutie_head_fwd / cutie_head_bwd can take it: 16-bit engine, 1x1, ReLU, at most 32 classes,
        feeding the loss softmax and nothing else; else None (f32 parity path, GHM's sigmoid logits)"""
        net = self.net
        lname = getattr(net, "loss_logits_name", None)
        if not self.head_fused or self.dt == F32 or lname is None or lname not in net.layers:
            return None
        n = net.layers[lname]
        a = n.attrs
        if n.op != "conv" or a["kh"] != 1 or a["kw"] != 1 or a.get("activation", "relu") != "relu" or n.channels > 32:
            return None
        cin_pad = _pad(a["cin"], self.bke)
        if cin_pad > 256 or n.domain != "grid":
            return None
        cons = [c for m in net.nodes for c in [m] if any(i.name == lname for i in m.inputs)]
        if len(cons) != 1 or cons[0].op != "softmax" or (getattr(net, "loss_spec", None) or {}).get("use_ghm"):
            return None
        return n

    def _plan_backward(self, P):
        if P.bwd_ready:
            return
        net, dev, tdt, B = self.net, self.dev, self.tdt, P.B
        for n in P.convs:
            hs, ws = P.spatial(n)
            P.dz[n.name] = torch.zeros((B, hs, ws, _pad(n.channels)), dtype=tdt, device=dev)
            t = ops.in_tiles(B, hs * ws, n.channels)       # H*W slices per sample of the two backward reductions
            # (sum dy, sum dy*xhat) slots: filled by the data gradient's own epilogue (one slot per conv tile;
            # the rest stay zero) or by in_bwd_stats (all of them)
            ct = ops.conv_stats_tiles(hs, ws)
            P.red[n.name] = torch.zeros((B, ct, n.channels, 2), dtype=torch.float32, device=dev)
            # the one-launch data gradient (conv_dgrad_in_bwd) leaves one bias-gradient row per conv tile
            fusable = n.name != net.loss_logits_name and self.dt != F32
            rows = max(ct, t) if fusable else t
            if n is P.head:
                rows = ops.head_chunks(hs * ws)          # cutie_head_bwd: one row per (sample, 256-cell chunk)
            P.dbpart[n.name] = torch.zeros((B * rows, n.channels), dtype=torch.float32, device=dev)
            # tagged granules of that launch's tile rendezvous (word 0: timeouts); zeroed once, then the library's
            P.sync[n.name] = torch.zeros(ops.conv_sync_bytes(B, ct, n.channels) // 4, dtype=torch.int32, device=dev)
        # gradient buffers for every tensor that a conv, the embedding or the sampler produces
        P.ps_acc = {}
        for n in net.nodes:
            if n.op == "conv":
                hs, ws = P.spatial(n)
                logits = n.name == net.loss_logits_name
                ch = _pad(n.channels, 4) if logits else _pad(n.channels)
                P.dy[n.name] = torch.zeros((B, hs, ws, ch), dtype=torch.float32 if logits else tdt, device=dev)
            elif n.op in ("embed", "positional_sampling"):
                P.dy[n.name] = torch.zeros((B, P.H, P.W, _pad(n.channels)), dtype=tdt, device=dev)
            if n.op == "positional_sampling":
                # f32 accumulator of the scatter-add into the sampled feature map (image domain)
                P.ps_acc[n.name] = torch.zeros((B, P.Hi * P.W, n.channels), dtype=torch.float32, device=dev)
        ws_bytes = 0
        for n in P.convs:
            a = n.attrs
            hs, ws = P.spatial(n)
            ws_bytes = max(ws_bytes, ops.conv_wgrad_workspace_bytes(B, hs, ws, a["cin"], n.channels, a["kh"], a["kw"]))
        P.wgrad_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        P.scratch = torch.zeros(capi.REDUCE_SCRATCH, dtype=torch.float32, device=dev)
        P.hnm_ws = torch.empty(ops.hnm_workspace_bytes(B * P.H * P.W), dtype=torch.uint8, device=dev)
        P.reg = torch.zeros(1, dtype=torch.float32, device=dev)
        P.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        P.bwd_ready = True

    def wgrad_flush_points(self):
        """Offsets of the flat gradient buffer at which loss_backward(on_ready=...) announces a finished prefix: the end
        of each of its batched weight-gradient calls (the first one closes when half of the conv range is pending)."""
        pts, pend = [], 0
        for n in reversed(self.net.nodes):
            if n.op != "conv":
                continue
            off, cnt, _ = self.offsets[n.name + "/weights"]
            pend += cnt
            if pend >= self.n_conv_w // 2:
                pts.append(_pad(off + cnt, 64))
                pend = 0
        if not pts or pts[-1] != self.n_conv_w:
            pts.append(self.n_conv_w)
        return pts

    def _grad_segments(self, P, name):
        """(dz view, packed dgrad weights, row offset) for every conv that consumes tensor `name`,
        directly or through a concat.  global_pool consumers contribute nothing: the pooled
        tensor is instance-normalised, so d(mean)/dx is annihilated by the norm's backward
        ('global pool gradient')."""
        segs = []
        self._walk_consumers(P, name, 0, segs)
        return segs

    def _walk_consumers(self, P, nm, off, segs):
        # a method, not a recursive local function: that would be a reference cycle (function <-> its own closure cell)
        # holding P - and an evicted plan's gigabytes - until the next gc pass
        for c in P.consumers.get(nm, []):
            if c.op == "conv":
                a = c.attrs
                segs.append(dict(x=P.dz[c.name], w=self.w_dg[c.name + "/weights"], cin=_pad(c.channels, self.bke),
                                 kh=a["kh"], kw=a["kw"], dil=a["rate"], flip=True, n_off=off))
            elif c.op == "concat":
                o = 0
                for i in c.inputs:
                    if i.name == nm:
                        break
                    o += i.channels
                self._walk_consumers(P, c.name, off + o, segs)
            elif c.op == "global_pool":
                # dropping this gradient is exact only behind an instance norm ('global pool gradient')
                if self.net.layers[nm].op != "conv":
                    raise NotImplementedError("gradient of global_pool(%s): the pooled tensor is not the output of an "
                                              "instance-normalised conv, so the pool's gradient does not vanish" % nm)
            elif c.op in ("softmax", "positional_sampling"):
                pass        # sampler: its scatter-add is added by loss_backward (_sampler_consumers)
            else:
                raise NotImplementedError("gradient through %s (%s)" % (c.name, c.op))

    # CUTIE_WGRAD_BATCH=0: every layer's weight gradient is its own call, right after the layer's dz (round-1 behaviour)
    wgrad_batch = os.environ.get("CUTIE_WGRAD_BATCH", "1") != "0"

    # CUTIE_TRAIN_FWD_FUSED=0: training keeps the pre-norm activation (conv, then in_apply), as inference did before
    train_fwd_fused = os.environ.get("CUTIE_TRAIN_FWD_FUSED", "1") != "0"

    def _act_operand(self, P, n):
        """what the backward kernels read as layer n's forward activation: the pre-norm activation a, or - where the
        forward launch stored the normalised output only - y itself (xhat = y, ReLU pattern inside y)"""
        if P.fused.get(n.name):
            return P.view(n.name, n.channels), True
        return P.a[n.name][..., :n.channels], False

    def relu_mask(self, P, name):
        """bool [B,H,W,C]: where layer `name`'s activation is non-zero in the last forward pass (the pattern the
        backward pass uses).  From a where it was stored; where only y was, from y's encoding: the a == 0 elements of a
        (sample, channel) plane all hold the plane's smallest value y0 = round16((0 - mean) * inv)."""
        n = self.net.layers[name]
        if n is P.head:
            return P.head_a[..., :n.channels] > 0
        if not P.fused.get(name):
            return P.a[name][..., :n.channels] > 0
        y = P.view(name, n.channels).float()
        st = P.stats[name].double().sum(dim=1)                               # [B, C, 2], about shift = act(bias)
        hw = y.shape[1] * y.shape[2]
        bias = self.params[name + "/biases"].double()
        act = n.attrs.get("activation", "relu")
        shift = torch.relu(bias) if act == "relu" else bias
        md = st[..., 0] / hw
        mean = md + shift
        inv = torch.rsqrt((st[..., 1] / hw - md * md).clamp_min(0) + ops.IN_EPS)
        y0 = ((0.0 - mean) * inv).float()[:, None, None, :]                  # the device rounds this to 16 bits
        ymin = y.amin(dim=(1, 2), keepdim=True)
        ulp = 2.0 ** (-10 if self.dt == F16 else -7)
        has_zero = (ymin - y0).abs() <= 2 * ulp * (y0.abs() + 1e-3)          # the plane's minimum IS y0: some a == 0
        return ~((y == ymin) & has_zero)

    def _sampler_consumers(self, P, name):
        """positional_sampling nodes that gather from tensor `name` (their input 0)"""
        return [c for c in P.consumers.get(name, []) if c.op == "positional_sampling" and c.inputs[0].name == name]

    def _backprop_into(self, P, n):
        """dy of node n's output from everything that consumes it.  For a conv node fed by convs the call goes all
        the way to dz (data gradient + the gradient of the node's instance norm and activation, one launch where
        the tiles can meet): returns True then, and loss_backward skips in_bwd_stats / in_bwd_apply."""
        segs = self._grad_segments(P, n.name)
        samplers = self._sampler_consumers(P, n.name)
        if samplers:
            # tf.batch_gather's gradient: scatter-add of the sampler's output gradient (model_framework.py:73-86)
            if segs or len(samplers) > 1:
                raise NotImplementedError("%s feeds a sampler and other layers: mixed gradient not built" % n.name)
            ps = samplers[0]
            acc = P.ps_acc[ps.name]
            acc.zero_()
            C = ps.channels
            dps = P.dy[ps.name]
            ops.batch_scatter_add(dps.as_strided((P.B, P.H * P.W, C), (dps.stride(0), dps.stride(2), 1)), P.ps, acc)
            ops.convert(acc.view(P.B, P.Hi, P.W, C), P.dy[n.name][..., :C])
        elif segs:
            if n.op == "conv":
                av, from_y = self._act_operand(P, n)
                ops.conv_dgrad_in_bwd(segs, P.dy[n.name][..., :n.channels], n.channels, av,
                                      P.stats[n.name], P.red[n.name], P.dz[n.name][..., :n.channels],
                                      dbias_part=P.dbpart[n.name], relu=n.attrs.get("activation", "relu"),
                                      sync=P.sync[n.name], bias=self.params[n.name + "/biases"], from_y=from_y,
                                      late=P.late)
                return True
            ops.conv_fwd(segs, P.dy[n.name], n.channels)
        return False

    # ------------------------------------------------------------------ execution
    def forward(self, grid, image=None, ps_1d_indices=None, labels=None):
        """grid int [B,H,W,1] or [B,H,W] (host or device).  Returns the plan holding every
        named layer; `softmax` and the per-cell CE are computed when the graph has a softmax.
        Without labels the pass is inference: the 16-bit paths then let a conv store its normalised output directly
        (no pre-norm activation is kept, loss_backward is not possible); each value is rounded once from f32 there and
        twice in the training form, so the two modes agree to a unit in the last place, not bit for bit."""
        net = self.net
        g = torch.as_tensor(grid)
        if g.dim() == 4:
            g = g[..., 0]
        B, H, W = (int(v) for v in g.shape)
        Hi = 0
        if image is not None:
            img = torch.as_tensor(image)
            Hi = int(img.shape[1])
            if int(img.shape[2]) != W or int(img.shape[0]) != B:
                raise CutieError("data_image %s does not match grid %s (width = cols)" % (tuple(img.shape), (B, H, W)))
        P = self._plan(B, H, W, Hi)
        # Device tensors of the right type are used IN PLACE, not copied (cutie_grid_build's int32 grid and int8 label grid
        # go straight into the kernels): the caller must leave them alone until the step that reads them is done.
        if g.is_cuda and g.device == self.dev and g.dtype == torch.int32 and g.is_contiguous():
            P.grid = g
        else:
            P.grid = P.grid_buf
            P.grid.copy_(g.to(torch.int32), non_blocking=True)
        if labels is not None:
            lab = torch.as_tensor(labels).reshape(B, H, W)
            in_place = lab.is_cuda and lab.device == self.dev and lab.is_contiguous() and (
                lab.dtype == torch.int32 or (lab.dtype == torch.int8 and P.head is not None))     # the head kernels take int8
            if in_place:
                P.labels = lab
            else:
                P.labels = P.labels_buf
                P.labels.copy_(lab.to(torch.int32), non_blocking=True)
        if Hi:
            P.bufs["data_image"][..., :3].copy_(img.to(self.dev))
            P.ps.copy_(torch.as_tensor(ps_1d_indices).to(torch.int32).reshape(B, H * W))
        # P.stats_all needs no clearing: every launch stores the same slots of it again, the others stay zero
        P.counters.zero_()
        for n in net.nodes:
            if n.op == "embed":
                ops.embed_fwd(P.grid, self.params[n.name + "/weights"], P.view(n.name, n.channels), P.oob)
            elif n.op == "conv" and n is P.head:
                # the logits layer, its norm, the softmax and (training) the CE + level 0 of the hard-negative select
                a = n.attrs
                src = n.inputs[0]
                cin_pad = _pad(a["cin"], self.bke)
                w = self.params[n.name + "/weights"]
                if cin_pad != a["cin"]:
                    raise NotImplementedError("logits head with cin %d not a multiple of %d" % (a["cin"], self.bke))
                ops.head_fwd(P.view(src.name, cin_pad), cin_pad, w, self.params[n.name + "/biases"], self.C, P.head_a,
                             P.head_stats, P.bufs[n.name], probs=P.probs.view(-1, self.C),
                             labels=P.labels if labels is not None else None, ce=P.ce, bad=P.bad, workspace=P.head_ws)
                P.fused[n.name] = False
            elif n.op == "conv":
                a = n.attrs
                src = n.inputs[0]
                cin_pad = _pad(a["cin"], self.bke)
                seg = [dict(x=P.view(src.name, cin_pad), w=self.w_fwd[n.name + "/weights"], cin=cin_pad,
                            kh=a["kh"], kw=a["kw"], dil=a["rate"])]
                bias, act = self.params[n.name + "/biases"], a.get("activation", "relu")
                if self.dt != F32 and P.view(n.name, n.channels).dtype == self.tdt and (
                        labels is None or (act != "sigmoid" and self.train_fwd_fused)):
                    # the conv may store the normalised output itself (a sample's tiles meet inside the launch;
                    # cutie_conv_in_fwd): no in_apply pass, and no pre-norm activation - the backward pass then takes
                    # xhat = y and reads the ReLU pattern out of y (P.fused)
                    if n.name not in P.sync_fw:
                        P.sync_fw[n.name] = torch.zeros(ops.conv_sync_bytes(B, P.stats[n.name].shape[1], n.channels) // 4,
                                                        dtype=torch.int32, device=self.dev)
                    P.fused[n.name] = ops.conv_in_fwd(seg, P.a[n.name][..., :n.channels], P.view(n.name, n.channels),
                                                      n.channels, bias, act, P.stats[n.name], sync=P.sync_fw[n.name],
                                                      late=P.late)
                else:
                    P.fused[n.name] = False
                    ops.conv_fwd(seg, P.a[n.name], n.channels, bias=bias, relu=act, stats=P.stats[n.name])
                    ops.in_apply(P.a[n.name][..., :n.channels], P.stats[n.name], P.view(n.name, n.channels),
                                 bias=bias, relu=act)
            elif n.op == "global_pool":
                src = n.inputs[0]
                ops.spatial_mean_bcast(P.view(src.name, src.channels), P.view(n.name, n.channels), P.pool_scratch)
            elif n.op == "concat":
                for (src, dst_name, dst) in P.copies:
                    if dst_name == n.name:
                        ops.convert(P.view(src, dst.shape[-1]), dst)
            elif n.op == "positional_sampling":
                feat = n.inputs[0]
                fv = P.view(feat.name, feat.channels)
                ops.batch_gather(fv.reshape(B, Hi * W, feat.channels) if fv.is_contiguous() else
                                 fv.as_strided((B, Hi * W, feat.channels), (fv.stride(0), fv.stride(2), 1)),
                                 P.ps, P.view(n.name, n.channels).as_strided(
                                     (B, H * W, n.channels),
                                     (P.view(n.name, n.channels).stride(0), P.view(n.name, n.channels).stride(2), 1)),
                                 oob=P.ps_oob)
            elif n.op == "softmax" and P.head is not None and n.inputs[0] is P.head:
                pass        # done by cutie_head_fwd
            elif n.op == "softmax":
                src = n.inputs[0]
                lab = P.labels if labels is not None else P.zero_labels
                ops.softmax_ce_fwd(P.view(src.name, src.channels), lab, src.channels, P.ce, P.probs, P.bad)
            else:
                raise NotImplementedError(n.op)
        P.has_labels = labels is not None
        return P

    def get(self, P, name):
        """Device tensor of a named layer after forward()."""
        n = self.net.layers[name]
        if n.op == "softmax":
            return P.probs
        if n.op == "input":
            return {"data_grid": P.grid, "gt_classes": P.labels}.get(name)
        return P.view(name, n.channels)

    def check_inputs(self, P):
        """Host-side validation of the counters the kernels keep (ONE device-to-host copy).  Call it after forward
        (input errors) and / or after the step: the late-tile counter covers every conv launch since the forward began,
        the backward ones included.  The device does not depend on this call: a step with late tiles is skipped by the
        optimizer kernel itself and counted in opt_state[4] (steps_invalid())."""
        cnt = P.counters.cpu()
        oob, bad, ps_oob, late = int(cnt[0]), int(cnt[1]) if P.has_labels else 0, int(cnt[2]), int(cnt[3])
        if ps_oob:
            raise CutieError("%d ps_1d_indices fall outside the image feature map (tf.batch_gather raises, "
                             "model_framework.py:83)" % ps_oob)
        if late:
            # the tiles' tags may have drifted apart: start every area of this plan from zero again
            for t in list(P.sync.values()) + list(P.sync_fw.values()):
                t.zero_()
            P.late.zero_()
            raise CutieError("%d conv waves gave up waiting for their sample's other tiles (the GPU could not hold a "
                             "sample's workgroups at once - a shared or partitioned device?): the results of this "
                             "step are invalid and the optimizer skipped it; set CUTIE_RENDEZVOUS=0 before loading "
                             "the library to run without the in-launch tile meeting" % late)
        if oob:
            raise CutieError("%d grid cells hold a vocabulary id outside [0, %d) (SURVEY.md 0.7)" % (oob, self.net.num_vocabs))
        if bad:
            raise CutieError("%d labels fall outside [0, %d) (entity_class_id overflow, SURVEY.md 0.6)" % (bad, self.C))

    def steps_invalid(self):
        """steps the device skipped because a conv tile's bounded wait ran out (one sync)"""
        return int(self.opt_state[4].item())

    def loss_backward(self, P, on_ready=None, defer_l2=None):
        """model_cutie.py:107-140 + tf.gradients: fills grads_flat (incl. the L2 term) and the
        squared global norm the way tf.clip_by_global_norm sees it.  Needs forward(labels=...).

        on_ready(upto): data-parallel hook.  The flat gradient buffer is laid out in backward
        order, so after each conv's weight gradient the prefix [0, upto) is final and the caller
        starts that slice's all-reduce while earlier layers still compute.  With the hook the conv
        range is left WITHOUT its L2 term (and, f16, still loss-scaled): the term is the same on
        every rank, so it is added once after the reduce - add_l2_after_reduce() - instead of by
        fourteen small launches inside backward.

        defer_l2 (default: with the hook only): leave grads_flat exactly as the gradient kernels wrote it - no L2 term,
        f16 still loss-scaled - for apply_update(), which takes both in its own two passes."""
        if not P.has_labels:
            raise CutieError("loss_backward needs forward(..., labels=...)")
        self._plan_backward(P)
        net, spec = self.net, self.net.loss_spec
        lname = net.loss_logits_name
        C = self.C
        ls = self.loss_scale
        if spec.get("use_ghm"):
            # model_cutie.py:110: the CE is taken of cls_logits * ghm_weights; the weights are a constant
            # input computed from this very forward pass (the reference's two partial_run phases)
            if getattr(self, "ghm_state", None) is None:
                self.ghm_state = ops.ghm_state(spec["ghm_bins"], self.dev)
            if getattr(P, "ghm_w", None) is None:
                P.ghm_w = torch.zeros((P.N, C), dtype=torch.float32, device=self.dev)
                P.ghm_z = torch.zeros((P.N, C), dtype=torch.float32, device=self.dev)
                P.ghm_probs = torch.zeros((P.N, C), dtype=torch.float32, device=self.dev)
            ops.ghm_weights(P.view(lname, C), P.labels, C, spec["ghm_bins"], spec["ghm_momentum"], spec["ghm_eps"],
                            self.ghm_state, P.ghm_w, P.ghm_z)
            P.bad.zero_()
            ops.softmax_ce_fwd(P.ghm_z, P.labels, C, P.ce, P.ghm_probs, P.bad)
            ops.hnm_weights(P.ce, P.labels, int(spec["hard_negative_ratio"]), P.weight, P.loss, P.hnm_ws)
            ops.softmax_ce_bwd(P.ghm_probs, P.labels, P.weight, C, P.dy[lname][..., :C], ls, P.ghm_w)
        elif P.head is not None:
            # the rest of the select, d(loss)/d(logits), and the logits layer's norm + ReLU gradient: dz, bias partials, loss
            ops.head_bwd(P.head_a, P.head_stats, P.bufs[lname], P.probs.view(-1, C), P.labels, P.ce, C,
                         self.params[lname + "/biases"], int(spec["hard_negative_ratio"]), P.weight, P.head_red, P.loss,
                         P.dz[lname], P.dbpart[lname], P.head_ws, loss_scale=ls)
        else:
            ops.hnm_weights(P.ce, P.labels, int(spec["hard_negative_ratio"]), P.weight, P.loss, P.hnm_ws)
            ops.softmax_ce_bwd(P.probs, P.labels, P.weight, C, P.dy[lname][..., :C], ls)
        if defer_l2 is None:
            defer_l2 = on_ready is not None
        # conv kernels and biases are overwritten by their gradient kernels (the padding between tensors is
        # never written and stays zero); only the embedding gradient is accumulated into - apply_update() leaves it cleared
        if not getattr(self, "_embed_grad_clean", False):
            self.grads_flat[self.n_dense:].zero_()
        self._embed_grad_clean = False
        if not defer_l2:
            P.reg.zero_()
        wd = float(spec["weight_decay"])
        main = torch.cuda.current_stream(self.dev)
        side = None
        if self.overlap_wgrad:
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            side = self._side
        # The weight gradients go out as batched calls (cutie_conv_wgrad_batched).  Data parallel (on_ready): TWO of them -
        # the layers of the first half of the conv range (the decoder and the ASPP side: the front of the flat buffer) as
        # soon as their dz exist, so that their buckets' all-reduce runs under the rest of backward, the others at the end.
        # (Round 2 kept layer-by-layer calls there because batches of 3-5 layers were 1.8-3 % slower than single calls;
        # two batches of ~7 fill the chip like the one of 14.)
        defer = self.wgrad_batch and side is None
        pending = []
        pending_elems = [0]

        def flush():
            if not pending:
                return
            key = tuple(m.name for m in pending)
            if getattr(P, "_wgrad_batches", None) is None:
                P._wgrad_batches = {}
            if key not in P._wgrad_batches:
                tasks = [(P.view(m.inputs[0].name, m.attrs["cin"]), m.attrs["cin"], P.dz[m.name][..., :m.channels], m.channels,
                          m.attrs["kh"], m.attrs["kw"], m.attrs["rate"], self.grads[m.name + "/weights"], False) for m in pending]
                chunks = [tasks[i:i + capi.MAX_BATCH] for i in range(0, len(tasks), capi.MAX_BATCH)]
                P._wgrad_batches[key] = [ops.wgrad_descs(c) for c in chunks]
            for arr, dtc in P._wgrad_batches[key]:
                need = ops.conv_wgrad_batched_workspace_bytes(arr, dtc)
                # one slab workspace for the engine, not one per grid shape (launches of all plans are in stream order)
                if getattr(self, "_wgrad_bws", None) is None or self._wgrad_bws.numel() < need:
                    self._wgrad_bws = None
                    self._wgrad_bws = torch.empty(need, dtype=torch.uint8, device=self.dev)
                ops.conv_wgrad_batched(arr, dtc, self._wgrad_bws)
            if on_ready is not None:
                off, cnt, _ = self.offsets[pending[-1].name + "/weights"]
                on_ready(_pad(off + cnt, 64))      # everything up to the last layer of the batch is final
            del pending[:]
            pending_elems[0] = 0

        for n in reversed(net.nodes):
            if n.op == "conv":
                a = n.attrs
                have_dz = self._backprop_into(P, n) if n.name != lname else (n is P.head)
                if not have_dz:
                    dy = P.dy[n.name][..., :n.channels]
                    av, from_y = self._act_operand(P, n)
                    bias, act = self.params[n.name + "/biases"], a.get("activation", "relu")
                    ops.in_bwd_stats(dy, av, P.stats[n.name], P.red[n.name], bias=bias, relu=act, from_y=from_y)
                    ops.in_bwd_apply(dy, av, P.stats[n.name], P.red[n.name], P.dz[n.name][..., :n.channels],
                                     dbias_part=P.dbpart[n.name], relu=act, bias=bias, from_y=from_y)
                src = n.inputs[0]
                if side is not None:
                    # dz of this layer is final: its weight gradient (and, data-parallel, the L2 term and the
                    # bucket hand-off) goes to the side stream, in layer order, sharing one workspace
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                if defer:
                    # the weight gradient only needs dz of this layer and the layer's input, both of which stay in place
                    # until the end of the pass: all layers' weight gradients go out as ONE call at the end (fewer K
                    # splits, one reduction launch - cutie_conv_wgrad_batched)
                    pending.append(n)
                    pending_elems[0] += self.offsets[n.name + "/weights"][1]
                    if on_ready is not None and pending_elems[0] >= self.n_conv_w // 2:
                        flush()
                    continue
                with torch.cuda.stream(side if side is not None else main):
                    ops.conv_wgrad(P.view(src.name, a["cin"]), a["cin"], P.dz[n.name][..., :n.channels], n.channels,
                                   a["kh"], a["kw"], a["rate"], self.grads[n.name + "/weights"], P.wgrad_ws)
                    if on_ready is not None:
                        off, cnt, _ = self.offsets[n.name + "/weights"]
                        on_ready(_pad(off + cnt, 64))
            elif n.op == "positional_sampling":
                self._backprop_into(P, n)
            elif n.op == "embed":
                self._backprop_into(P, n)
                if getattr(P, "embed_ws", None) is None:
                    P.embed_ws = torch.empty(ops.embed_bwd_workspace_bytes(P.grid.numel(), n.channels), dtype=torch.uint8,
                                             device=self.dev)
                ops.embed_bwd(P.grid, P.dy[n.name][..., :n.channels], self.grads[n.name + "/weights"], P.embed_ws)
                P.d_embed = P.dy[n.name]
        flush()
        if side is not None:
            main.wait_stream(side)     # every weight gradient is in place; the buffers it read may be reused
        # bias gradients: fold the per-sample partials of every layer in one launch
        if getattr(P, "_sum_rows_batches", None) is None:
            # partial rows per layer differ (B x slices of H*W): group the layers by row count
            groups = OrderedDict()
            for n in P.convs:
                groups.setdefault(P.dbpart[n.name].shape[0], []).append((P.dbpart[n.name], self.grads[n.name + "/biases"]))
            P._sum_rows_batches = [(rows, ops.sum_rows_batched(ents, rows)) for rows, ents in groups.items()]
        else:
            for rows, batches in P._sum_rows_batches:
                ops.replay_sum_rows(batches, rows, self.dev)
        P.reg_div = 1.0
        P.l2_pending = bool(defer_l2)
        P.raw_grads = bool(defer_l2)       # apply_update() wants them so; add_l2_after_reduce() finishes them in place
        if not defer_l2:
            ops.l2_add(self.params_flat[:self.n_conv_w], self.grads_flat[:self.n_conv_w], wd, P.reg, P.scratch, ls)
        if ls is not None and not defer_l2:
            # biases and the embedding carry no L2 term: wd = 0 only takes the loss scale out
            ops.l2_add(self.params_flat[self.n_conv_w:], self.grads_flat[self.n_conv_w:], 0.0, P.reg, P.scratch, ls)
        return P

    def apply_update(self, P, m, v, lr_t, beta1, beta2, eps, clip, world=1, t=0, emb_rows_sq=None,
                     scale_growth_interval=1000):
        """The optimizer tail of main_train_json.py:152-156 after loss_backward(defer_l2=True) (and, data parallel, the
        all-reduce): L2 term, tf.clip_by_global_norm, Adam, the repack of the conv operands and the loss-scale update in
        three launches (cutie_opt_norm, cutie_opt_step, cutie_loss_scale_update).  grads_flat holds the SUM of `world`
        replicas' raw gradients.  emb_rows_sq: the embedding rows' share of the squared norm summed over the ranks
        (Trainer.step); None = this replica's, taken inside the first pass.  A step whose norm is not finite (f16
        overflow) or whose conv launches counted a late tile (P.late) changes nothing and is counted in opt_state.
        Returns the 1-float device tensor of the squared global norm (of the summed gradient)."""
        if not getattr(P, "raw_grads", False):
            raise CutieError("apply_update needs loss_backward(..., defer_l2=True)")
        wd = float(self.net.loss_spec["weight_decay"])
        if getattr(self, "_opt_layers", None) is None:
            names = sorted(self.conv_w_names, key=lambda k: self.offsets[k][0])
            self._opt_layers = ops.opt_layers([(self.params[k], self.w_fwd[k], self.w_dg[k]) for k in names])
            self._opt_partials = torch.zeros(capi.OPT_SCRATCH, dtype=torch.float32, device=self.dev)
        rows = None
        if emb_rows_sq is None and hasattr(P, "d_embed"):
            rows = P.d_embed.reshape(-1)
        ops.opt_norm(self.params_flat, self.grads_flat, self.n_conv_w, self.n_dense, wd * world, self._opt_partials,
                     rows=rows, loss_scale=self.loss_scale)
        ops.opt_step(self.params_flat, self.grads_flat, m, v, self.n_conv_w, wd * world, 0.5 * wd, self._opt_layers, self.dt,
                     lr_t, beta1, beta2, eps, clip, 1.0 / world, t, self._opt_partials, extra_sq=emb_rows_sq,
                     loss_scale=self.opt_state, invalid=P.late, gnorm_sq_out=P.gnorm_sq, reg_out=P.reg,
                     zero_from=self.n_dense)
        f16 = self.loss_scale is not None
        ops.loss_scale_update(self.opt_state, P.gnorm_sq, backoff=0.5 if f16 else 1.0, growth=2.0 if f16 else 1.0,
                              interval=scale_growth_interval, invalid=P.late)
        P.reg_div, P.l2_pending, P.raw_grads = 1.0, False, False
        self._embed_grad_clean = True      # cutie_opt_step cleared the embedding gradient behind itself
        return P.gnorm_sq

    def add_l2_after_reduce(self, P, world=1):
        """Data-parallel tail of loss_backward(on_ready=...): the conv range holds the SUM of the ranks' gradients;
        the optimizer applies 1/world to everything, so the L2 term goes in as world * wd * w (and, f16, the loss
        scale comes out) - mean gradient + wd * w, what every rank adding its own term before the reduce gives."""
        if not getattr(P, "l2_pending", False):
            return
        wd = float(self.net.loss_spec["weight_decay"])
        P.reg.zero_()
        ops.l2_add(self.params_flat[:self.n_conv_w], self.grads_flat[:self.n_conv_w], wd * world, P.reg, P.scratch,
                   self.loss_scale)
        if self.loss_scale is not None:
            # biases and the embedding carry no L2 term: wd = 0 only takes the loss scale out
            ops.l2_add(self.params_flat[self.n_conv_w:], self.grads_flat[self.n_conv_w:], 0.0, P.reg, P.scratch, self.loss_scale)
        P.reg_div = float(world)       # the kernel's 0.5 * (world * wd) * sum(w^2) is world times the loss term
        P.l2_pending = False
        P.raw_grads = False            # finished in place: cutie_adam_step's input, no longer apply_update()'s

    def embedding_rows_sumsq(self, P):
        """sum over this replica's grid cells of |d(embedding output) row|^2, loss scale taken out: the embedding's
        share of tf.clip_by_global_norm (IndexedSlices values, un-deduplicated).  A 1-float device tensor."""
        P.emb_sq = self.grads_extra      # rides behind the gradient buffer (GradBuckets.finish(tail_extra=1))
        P.emb_sq.zero_()
        if hasattr(P, "d_embed"):
            ops.sumsq(P.d_embed.reshape(-1), P.emb_sq, P.scratch, self.loss_scale)     # still scaled: activation dtype
        return P.emb_sq

    def global_norm_sq(self, P, tf_indexed_slices=True, embedding_rows_sq=None):
        """Squared global norm the way tf.clip_by_global_norm sees it: every dense gradient, and the
        embedding through its un-deduplicated per-cell rows.  embedding_rows_sq: that share already summed over the
        data-parallel ranks (Trainer.step); tf_indexed_slices=False takes the dense summed embedding gradient
        instead (not TF's definition; kept for comparisons)."""
        P.gnorm_sq.zero_()
        if embedding_rows_sq is not None:
            ops.sumsq(self.grads_flat[:self.n_dense], P.gnorm_sq, P.scratch)
            P.gnorm_sq.add_(embedding_rows_sq)
        elif tf_indexed_slices and hasattr(P, "d_embed"):
            ops.sumsq(self.grads_flat[:self.n_dense], P.gnorm_sq, P.scratch)
            ops.sumsq(P.d_embed.reshape(-1), P.gnorm_sq, P.scratch, self.loss_scale)   # still scaled: activation dtype
        else:
            ops.sumsq(self.grads_flat, P.gnorm_sq, P.scratch)
        return P.gnorm_sq

    def loss_scale_value(self):
        """(S, steps skipped so far) - one sync; (1.0, 0) when the engine runs without a loss scale"""
        h = self.opt_state.cpu()
        return float(h[0]), int(h[3])

    def losses(self, P):
        """(model_loss, regularization_loss, total_loss) as python floats (one sync)."""
        ml, rg = float(P.loss[0].item()), float(P.reg.item()) / getattr(P, "reg_div", 1.0)
        return ml, rg, ml + rg
