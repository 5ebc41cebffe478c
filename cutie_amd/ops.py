"""Tensor-level wrappers over the C ABI: argument checking and pointer plumbing only.

Every function launches HIP kernels on the current stream and returns device tensors.
Activations are NHWC tensors (possibly channel-slice views of a wider concat buffer).
"""
import ctypes

import torch

from . import capi
from .capi import BF16, F32, ConvSeg, call, dt_of, ld_of, ptr, stream

IN_EPS = 1e-6  # tf.contrib.layers.instance_norm default epsilon (model_framework.py:423)
ACTIVATION_CODE = {None: 0, False: 0, "none": 0, True: 1, "relu": 1, "sigmoid": 2}


def _act(relu, from_y=False):
    """activation code; from_y: the `a` operand of a backward function is the layer's normalised output as the one-launch
    forward stores it (CUTIE_ACT_FROM_Y)"""
    return ACTIVATION_CODE[relu] | (capi.ACT_FROM_Y if from_y else 0)


def _bke(dt):
    return 32 if dt == F32 else 64


def pad_to(c, m):
    return (c + m - 1) // m * m


# ------------------------------------------------------------------------------------ K0
def grid_build(tok_box, tok_id, tok_ecid, doc_extent, doc_off, rows, cols, strategy):
    """device int32 tensors in; dict of device tensors out (see include/cutie_hip.h K0)."""
    dev = tok_box.device
    B = doc_off.numel() - 1
    T = tok_id.numel()
    grid = torch.empty((B, rows, cols), dtype=torch.int32, device=dev)
    labels = torch.empty((B, rows, cols), dtype=torch.int8, device=dev)
    ps_x = torch.empty((B, rows, cols), dtype=torch.int32, device=dev)
    ps_y = torch.empty((B, rows, cols), dtype=torch.int32, device=dev)
    # the kernel writes every entry of these three (-1 for the tokens of a rejected document)
    order = torch.empty((max(T, 1),), dtype=torch.int32, device=dev)
    place = torch.empty((max(T, 1), 2), dtype=torch.int32, device=dev)
    status = torch.empty((max(B, 1),), dtype=torch.int32, device=dev)
    ws_bytes = capi.lib().cutie_grid_build_workspace_bytes(T)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    call("cutie_grid_build", ptr(tok_box), ptr(tok_id), ptr(tok_ecid), ptr(doc_extent), ptr(doc_off), B, T,
         rows, cols, strategy, ptr(grid), ptr(labels), ptr(ps_x), ptr(ps_y), ptr(order), ptr(place),
         ptr(status), ptr(ws), ws_bytes, stream())
    return {"grid": grid, "labels": labels, "ps_x": ps_x, "ps_y": ps_y, "order": order[:T],
            "place": place[:T], "status": status[:B]}


# ------------------------------------------------------------------------------------ K1
def embed_fwd(ids, table, out, oob):
    """ids int32 [...]; table f32 [V,E]; out [..., >=E] view; oob int32 [1] counter."""
    V, E = table.shape
    call("cutie_embed_fwd", ptr(ids), ids.numel(), ptr(table), V, E, ptr(out), ld_of(out), dt_of(out),
         ptr(oob), stream())
    return out


def embed_bwd_workspace_bytes(n_cells, E):
    return capi.lib().cutie_embed_bwd_workspace_bytes(int(n_cells), int(E))


def embed_bwd(ids, dout, dtable, workspace=None):
    """workspace: optional uint8 tensor of embed_bwd_workspace_bytes(ids.numel(), E) bytes - with it the two hot rows
    (ids 0 and 1) are folded in a fixed order instead of by atomics"""
    V, E = dtable.shape
    call("cutie_embed_bwd", ptr(ids), ids.numel(), ptr(dout), ld_of(dout), dt_of(dout), ptr(dtable), V, E,
         ptr(workspace), workspace.numel() * workspace.element_size() if workspace is not None else 0, stream())
    return dtable


# ------------------------------------------------------------------------------------ K2-K5
def wgrad_descs(tasks):
    """tasks: list of (x view [B,H,W,>=cin], cin, dz view [B,H,W,>=cout], cout, kh, kw, dil, dw f32 [kh,kw,cin,cout],
    accumulate) -> (descriptor array, dtype code); all tasks share one element type"""
    if not 1 <= len(tasks) <= capi.MAX_BATCH:
        raise capi.CutieError("conv_wgrad_batched: %d layers (1..%d)" % (len(tasks), capi.MAX_BATCH))
    arr = (capi.WgradDesc * len(tasks))()
    dt = None
    for i, (x, cin, dz, cout, kh, kw, dil, dw, acc) in enumerate(tasks):
        if dt is None:
            dt = dt_of(x)
        if dt_of(x) != dt or dt_of(dz) != dt:
            raise capi.CutieError("conv_wgrad_batched: one element type per call")
        if dw.dtype != torch.float32 or not dw.is_contiguous() or tuple(dw.shape) != (kh, kw, cin, cout):
            raise capi.CutieError("conv_wgrad_batched: dw must be contiguous f32 [kh,kw,cin,cout]")
        if x.shape[:3] != dz.shape[:3]:
            raise capi.CutieError("conv_wgrad_batched: x and dz differ in [B,H,W]")
        d = arr[i]
        d.x, d.ld_x, d.cin, d.dz, d.ld_dz, d.cout = ptr(x), ld_of(x), cin, ptr(dz), ld_of(dz), cout
        d.B, d.H, d.W, d.kh, d.kw, d.dil = x.shape[0], x.shape[1], x.shape[2], kh, kw, dil
        d.dw, d.accumulate = ptr(dw), 1 if acc else 0
    return arr, dt


def conv_wgrad_batched_workspace_bytes(arr, dt):
    return capi.lib().cutie_conv_wgrad_batched_workspace_bytes(arr, len(arr), dt)


def conv_wgrad_batched(arr, dt, workspace):
    """the weight gradients of several layers in one call (cutie_conv_wgrad_batched): layers on the same kernel instance
    share one launch with fewer K splits; arr / dt from wgrad_descs (kept by the caller for replay)."""
    call("cutie_conv_wgrad_batched", arr, len(arr), dt, ptr(workspace), workspace.numel() * workspace.element_size(),
         stream(workspace.device))


def pack_weights(w_hwio, dt, want_fwd=True, want_dgrad=True):
    """f32 [kh,kw,cin,cout] -> (w_fwd [taps,cout,cin_pad], w_dgrad [taps,cin,cout_pad])."""
    kh, kw, cin, cout = w_hwio.shape
    tdt = capi.torch_dtype(dt)
    cin_pad, cout_pad = pad_to(cin, _bke(dt)), pad_to(cout, _bke(dt))
    w_fwd = torch.empty((kh * kw, cout, cin_pad), dtype=tdt, device=w_hwio.device) if want_fwd else None
    w_dg = torch.empty((kh * kw, cin, cout_pad), dtype=tdt, device=w_hwio.device) if want_dgrad else None
    pack_weights_into(w_hwio, dt, w_fwd, w_dg)
    return w_fwd, w_dg


def pack_weights_into(w_hwio, dt, w_fwd, w_dg):
    kh, kw, cin, cout = w_hwio.shape
    call("cutie_pack_weights", ptr(w_hwio), kh * kw, cin, cout, ptr(w_fwd),
         w_fwd.shape[2] if w_fwd is not None else 0, ptr(w_dg), w_dg.shape[2] if w_dg is not None else 0,
         dt, stream())


def pack_weights_batched(entries, dt):
    """entries: list of (w_hwio f32 [kh,kw,cin,cout], w_fwd or None, w_dgrad or None); one launch per
    capi.MAX_BATCH layers.  Returns the descriptor arrays so that a caller can keep and replay them."""
    batches = []
    for lo in range(0, len(entries), capi.MAX_BATCH):
        chunk = entries[lo:lo + capi.MAX_BATCH]
        arr = (capi.PackDesc * len(chunk))()
        for i, (w, wf, wd) in enumerate(chunk):
            kh, kw, cin, cout = w.shape
            if not w.is_contiguous() or w.dtype != torch.float32:
                raise capi.CutieError("pack_weights_batched: weights must be contiguous f32 HWIO")
            arr[i].w_hwio, arr[i].w_fwd, arr[i].w_dgrad = ptr(w), ptr(wf), ptr(wd)
            arr[i].taps, arr[i].cin, arr[i].cout = kh * kw, cin, cout
            arr[i].cin_pad = wf.shape[2] if wf is not None else 0
            arr[i].cout_pad = wd.shape[2] if wd is not None else 0
        batches.append(arr)
    replay_pack(batches, dt, entries[0][0].device)
    return batches


def replay_pack(batches, dt, device):
    for arr in batches:
        call("cutie_pack_weights_batched", arr, len(arr), dt, stream(device))


def sum_rows_batched(entries, B, accumulate=False):
    """entries: list of (part f32 [B,C], out f32 [C]).  Returns the descriptor arrays for replay."""
    batches = []
    for lo in range(0, len(entries), capi.MAX_BATCH):
        chunk = entries[lo:lo + capi.MAX_BATCH]
        arr = (capi.SumRowsDesc * len(chunk))()
        for i, (part, out) in enumerate(chunk):
            if part.shape[0] != B or part.shape[1] != out.numel() or not part.is_contiguous():
                raise capi.CutieError("sum_rows_batched: entry %d shapes %s -> %s" % (i, tuple(part.shape), tuple(out.shape)))
            arr[i].part, arr[i].out, arr[i].C = ptr(part), ptr(out), out.numel()
        batches.append(arr)
    replay_sum_rows(batches, B, entries[0][0].device, accumulate)
    return batches


def replay_sum_rows(batches, B, device, accumulate=False):
    for arr in batches:
        call("cutie_sum_rows_batched", arr, len(arr), B, 1 if accumulate else 0, stream(device))



def _seg_array(segs, out, n_out, what):
    B, H, W = out.shape[0], out.shape[1], out.shape[2]
    arr = (ConvSeg * len(segs))()
    dt = None
    for i, s in enumerate(segs):
        x, w = s["x"], s["w"]
        sdt = dt_of(x)
        if dt is None:
            dt = sdt
        if sdt != dt or dt_of(w) != dt:
            raise capi.CutieError("%s: segment dtypes differ" % what)
        if x.shape[0] != B or x.shape[1] != H or x.shape[2] != W:
            raise capi.CutieError("%s: segment %d spatial shape %s != out %s" % (what, i, tuple(x.shape), tuple(out.shape)))
        cin = s["cin"]
        if w.shape[2] != cin or x.shape[3] < cin and ld_of(x) < cin:
            raise capi.CutieError("%s: segment %d cin=%d, w %s, x %s" % (what, i, cin, tuple(w.shape), tuple(x.shape)))
        n_off = s.get("n_off", 0)
        if n_off + n_out > w.shape[1] or w.shape[0] != s["kh"] * s["kw"]:
            raise capi.CutieError("%s: segment %d weight rows %d < %d+%d" % (what, i, w.shape[1], n_off, n_out))
        arr[i].x = ptr(x)
        arr[i].w = ptr(w) + n_off * cin * w.element_size()
        arr[i].w_tap_stride = w.shape[1] * w.shape[2]
        arr[i].ld_x = ld_of(x)
        arr[i].cin = cin
        arr[i].kh, arr[i].kw, arr[i].dil = s["kh"], s["kw"], s["dil"]
        arr[i].flip = 1 if s.get("flip") else 0
    return arr, dt


def conv_sync_bytes(B, red_tiles, n_out):
    return capi.lib().cutie_conv_sync_bytes(int(B), int(red_tiles), int(n_out))


def conv_in_fwd(segs, a, y, n_out, bias, relu, stats, sync=None, eps=IN_EPS, late=None):
    """conv + bias + activation + instance norm as one call (cutie_conv_in_fwd): a = scratch view for the pre-norm
    activation, y = output view, stats as conv_fwd.  With `sync` (int32 tensor of conv_sync_bytes(B, T, n_out) bytes, zeroed
    once, dedicated to this layer and shape) the launch may store y directly and leave `a` unwritten; returns True when it
    did - the backward functions then take y in place of a with from_y=True.  late: the caller's step-wide int32 [1]
    counter of waits that ran out (see conv_dgrad_in_bwd)."""
    B = y.shape[0]
    arr, dt = _seg_array(segs, y, n_out, "conv_in_fwd")
    if dt_of(a) != dt or dt_of(y) != dt:
        raise capi.CutieError("conv_in_fwd: a and y must have the segments' dtype")
    fused = ctypes.c_int(0)
    call("cutie_conv_in_fwd", arr, len(segs), B, y.shape[1], y.shape[2], n_out, ptr(a), ld_of(a), ptr(y), ld_of(y), ptr(bias),
         ACTIVATION_CODE[relu], ptr(stats), _tiles(stats, B, n_out, True), eps, ptr(sync),
         sync.numel() * sync.element_size() if sync is not None else 0, ptr(late), ctypes.byref(fused), dt, stream())
    return bool(fused.value)


def conv_dgrad_in_bwd(segs, dy, n_out, a, stats, red, dz, dbias_part=None, relu=True, sync=None, eps=IN_EPS, bias=None,
                      from_y=False, late=None):
    """Data gradient of an instance-normalised layer and the gradient of its norm + activation in one call
    (cutie_conv_dgrad_in_bwd): segs as conv_fwd (flip=True), dy = scratch view [B,H,W,>=n_out], a / stats = the layer's
    forward activation and moments, red = f32 [B,T,n_out,2] zeroed once, dz = output view, dbias_part f32 [B*T', n_out],
    sync = int32 tensor of conv_sync_bytes(B, T, n_out) bytes zeroed once and dedicated to this layer and shape
    (None: two-call form).  sync[0] counts tiles whose bounded wait ran out.  bias: the layer's bias - with `relu` the
    shift its moments are kept about (None when they were produced without one).  late: int32 [1] device counter of the
    caller's, bumped together with sync[0]: one word for a whole step, which the optimizer entry points take as
    `invalid` and skip the update on."""
    B = dy.shape[0]
    arr, dt = _seg_array(segs, dy, n_out, "conv_dgrad_in_bwd")
    if dt_of(dy) != dt or dt_of(a) != dt or dt_of(dz) != dt:
        raise capi.CutieError("conv_dgrad_in_bwd: dy, a and dz must have the segments' dtype")
    call("cutie_conv_dgrad_in_bwd", arr, len(segs), B, dy.shape[1], dy.shape[2], n_out, ptr(dy), ld_of(dy), ptr(a), ld_of(a),
         ptr(stats), _tiles(stats, B, n_out, True), eps, ptr(red), _tiles(red, B, n_out, True), _act(relu, from_y),
         ptr(bias), ptr(dz), ld_of(dz), ptr(dbias_part), _tiles(dbias_part, B, n_out, False), ptr(sync),
         sync.numel() * sync.element_size() if sync is not None else 0, ptr(late), dt, stream())
    return dz


def conv_fwd(segs, out, n_out, bias=None, relu=False, stats=None, norm_bwd=None):
    """segs: list of dict(x=[B,H,W,>=cin] view, w=packed [taps, n_rows, cin], cin, kh, kw, dil, flip,
    n_off=row offset into w).  out: [B,H,W,>=n_out] view.  stats: zeroed f32 [B, T, n_out, 2] with
    T >= conv_stats_tiles(H, W): per-tile partial moments (sum over T for the totals).
    norm_bwd = (a, stats, red[, shift_bias, shift_act]) makes this a data-gradient launch that also leaves
    in_bwd_stats(out, a, stats) in `red` (f32 [B, T, n_out, 2], zeroed once by the caller; cutie_conv_dgrad_in_stats)."""
    B, H, W = out.shape[0], out.shape[1], out.shape[2]
    arr = (ConvSeg * len(segs))()
    dt = None
    for i, s in enumerate(segs):
        x, w = s["x"], s["w"]
        sdt = dt_of(x)
        if dt is None:
            dt = sdt
        if sdt != dt or dt_of(w) != dt:
            raise capi.CutieError("conv_fwd: segment dtypes differ")
        if x.shape[0] != B or x.shape[1] != H or x.shape[2] != W:
            raise capi.CutieError("conv_fwd: segment %d spatial shape %s != out %s" % (i, tuple(x.shape), tuple(out.shape)))
        cin = s["cin"]
        if w.shape[2] != cin or x.shape[3] < cin and ld_of(x) < cin:
            raise capi.CutieError("conv_fwd: segment %d cin=%d, w %s, x %s" % (i, cin, tuple(w.shape), tuple(x.shape)))
        n_off = s.get("n_off", 0)
        if n_off + n_out > w.shape[1] or w.shape[0] != s["kh"] * s["kw"]:
            raise capi.CutieError("conv_fwd: segment %d weight rows %d < %d+%d" % (i, w.shape[1], n_off, n_out))
        arr[i].x = ptr(x)
        arr[i].w = ptr(w) + n_off * cin * w.element_size()
        arr[i].w_tap_stride = w.shape[1] * w.shape[2]
        arr[i].ld_x = ld_of(x)
        arr[i].cin = cin
        arr[i].kh, arr[i].kw, arr[i].dil = s["kh"], s["kw"], s["dil"]
        arr[i].flip = 1 if s.get("flip") else 0
    if norm_bwd is not None:
        if bias is not None or ACTIVATION_CODE[relu] or stats is not None:
            raise capi.CutieError("conv_fwd: norm_bwd goes with a plain data gradient (no bias / activation / stats)")
        a, fstats, red = norm_bwd[:3]
        sb, sa = (norm_bwd[3], norm_bwd[4]) if len(norm_bwd) > 3 else (None, None)
        call("cutie_conv_dgrad_in_stats", arr, len(segs), B, H, W, n_out, ptr(out), ld_of(out), dt_of(out), ptr(a),
             ld_of(a), dt_of(a), ptr(fstats), _tiles(fstats, B, n_out, True), IN_EPS, ptr(red),
             _tiles(red, B, n_out, True), ptr(sb), ACTIVATION_CODE[sa], dt, stream())
        return out
    call("cutie_conv_fwd", arr, len(segs), B, H, W, n_out, ptr(out), ld_of(out), dt_of(out), ptr(bias),
         ACTIVATION_CODE[relu], ptr(stats), _tiles(stats, B, n_out, True), dt, stream())
    return out


def conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw):
    return capi.lib().cutie_conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw)


def conv_wgrad(x, cin, dz, cout, kh, kw, dil, dw, workspace, accumulate=False):
    """dw f32 [kh,kw,cin,cout] (HWIO); x/dz NHWC views; workspace uint8 tensor."""
    B, H, W = dz.shape[0], dz.shape[1], dz.shape[2]
    call("cutie_conv_wgrad", ptr(x), ld_of(x), cin, ptr(dz), ld_of(dz), cout, B, H, W, kh, kw, dil, ptr(dw),
         1 if accumulate else 0, dt_of(x), ptr(workspace), workspace.numel() * workspace.element_size(), stream())
    return dw


# ------------------------------------------------------------------------------------ K6, K8
def _bhwc(t):
    return t.shape[0], t.shape[1] * t.shape[2], t.shape[3]


def _tiles(t, B, C, last2):
    """slot count of a partial-sum tensor: [B, T, C, 2] / [B, C, 2] (T = 1) when last2, else [B*T, C]"""
    if t is None:
        return 0
    if not t.is_contiguous() or t.dtype != torch.float32:
        raise capi.CutieError("partial sums must be contiguous f32")
    per = C * (2 if last2 else 1)
    if t.numel() % (B * per):
        raise capi.CutieError("partial-sum tensor %s does not hold whole [B=%d][T][C=%d] slots" % (tuple(t.shape), B, C))
    return t.numel() // (B * per)


def conv_stats_tiles(H, W):
    return capi.lib().cutie_conv_stats_tiles(int(H), int(W))


def in_tiles(B, HW, C):
    return capi.lib().cutie_in_tiles(int(B), int(HW), int(C))


# The moments in `stats` are kept about shift[c] = act(bias[c]) (include/cutie_hip.h, K6): every function below takes the
# (bias, relu) pair the producer of `stats` used - conv_fwd uses its own bias and activation; bias=None means no shift.
def in_stats(a, stats, bias=None, relu=None):
    B, HW, C = _bhwc(a)
    call("cutie_in_stats", ptr(a), ld_of(a), dt_of(a), B, HW, C, ptr(stats), _tiles(stats, B, C, True), ptr(bias),
         ACTIVATION_CODE[relu], stream())
    return stats


def in_apply(a, stats, y, eps=IN_EPS, bias=None, relu=None):
    B, HW, C = _bhwc(a)
    call("cutie_in_apply", ptr(a), ld_of(a), dt_of(a), ptr(stats), _tiles(stats, B, C, True), eps, ptr(y), ld_of(y),
         dt_of(y), B, HW, C, ptr(bias), ACTIVATION_CODE[relu], stream())
    return y


def in_bwd_stats(dy, a, stats, red, eps=IN_EPS, bias=None, relu=None, from_y=False):
    B, HW, C = _bhwc(a)
    call("cutie_in_bwd_stats", ptr(dy), ld_of(dy), dt_of(dy), ptr(a), ld_of(a), dt_of(a), ptr(stats),
         _tiles(stats, B, C, True), eps, ptr(red), _tiles(red, B, C, True), B, HW, C, ptr(bias), _act(relu, from_y),
         stream())
    return red


def in_bwd_apply(dy, a, stats, red, dz, dbias_part=None, relu=True, eps=IN_EPS, bias=None, from_y=False):
    """dbias_part: f32 [B * T, C] partial sums, one row per (sample, slice of H*W) (fold with sum_rows)."""
    B, HW, C = _bhwc(a)
    call("cutie_in_bwd_apply", ptr(dy), ld_of(dy), dt_of(dy), ptr(a), ld_of(a), dt_of(a), ptr(stats),
         _tiles(stats, B, C, True), eps, ptr(red), _tiles(red, B, C, True), _act(relu, from_y), ptr(dz), ld_of(dz),
         dt_of(dz), ptr(dbias_part), _tiles(dbias_part, B, C, False), B, HW, C, ptr(bias), stream())
    return dz


def sum_rows(part, out, accumulate=False):
    B, C = part.shape
    call("cutie_sum_rows", ptr(part), B, C, ptr(out), 1 if accumulate else 0, stream())
    return out


def spatial_mean_bcast(x, out, scratch=None):
    """scratch: optional f32 tensor of >= B*C*2 elements (clobbered) - selects the wide-grid path"""
    B, HW, C = _bhwc(x)
    if scratch is not None and (scratch.dtype != torch.float32 or scratch.numel() < B * capi.POOL_TILES * C * 2):
        raise capi.CutieError("spatial_mean_bcast: scratch needs %d f32 elements" % (B * capi.POOL_TILES * C * 2))
    call("cutie_spatial_mean_bcast", ptr(x), ld_of(x), dt_of(x), ptr(out), ld_of(out), dt_of(out), B, HW, C,
         ptr(scratch), stream())
    return out


# ------------------------------------------------------------------------------------ K9-K11
def softmax_ce_fwd(logits, labels, C, ce, probs, bad):
    """logits [..., >=C] view; labels int32 [...]; ce f32 [N]; probs f32 [N,C] or None."""
    n = labels.numel()
    call("cutie_softmax_ce_fwd", ptr(logits), ld_of(logits), dt_of(logits), ptr(labels), n, C, ptr(ce),
         ptr(probs), ptr(bad), stream())


def hnm_workspace_bytes(n_cells):
    return capi.lib().cutie_hnm_workspace_bytes(n_cells)


def hnm_weights(ce, labels, ratio, weight, loss_out, workspace=None):
    if workspace is None:
        workspace = torch.empty(hnm_workspace_bytes(labels.numel()), dtype=torch.uint8, device=ce.device)
    call("cutie_hnm_weights", ptr(ce), ptr(labels), labels.numel(), int(ratio), ptr(weight), ptr(loss_out),
         ptr(workspace), workspace.numel(), stream())


def softmax_ce_bwd(probs, labels, weight, C, dlogits, loss_scale=None, elem_weight=None):
    call("cutie_softmax_ce_bwd", ptr(probs), ptr(labels), ptr(weight), labels.numel(), C, ptr(dlogits),
         ld_of(dlogits), dt_of(dlogits), ptr(loss_scale), ptr(elem_weight), stream())


def _label_bytes(labels):
    if labels.dtype == torch.int8:
        return 1
    if labels.dtype == torch.int32:
        return 4
    raise capi.CutieError("labels must be int8 or int32, not %s" % labels.dtype)


def head_tiles(HW):
    return capi.lib().cutie_head_tiles(int(HW))


def head_chunks(HW):
    return capi.lib().cutie_head_chunks(int(HW))


def head_workspace_bytes(B, HW):
    return capi.lib().cutie_head_workspace_bytes(int(B), int(HW))


def head_fwd(x, cin, w_hwio, bias, C, a, stats, logits, probs=None, labels=None, ce=None, bad=None, workspace=None, eps=IN_EPS):
    """the logits layer + instance norm + softmax (+ CE and level 0 of the hard-negative select with labels) - cutie_head_fwd.
    x [B,H,W,>=cin] 16-bit view; w_hwio f32 [1,1,cin,C] contiguous; a / logits f32 [B,H,W,ld]; stats f32 [B,T,C,2];
    probs f32 [B*H*W, C] contiguous; labels int8 / int32 [B,H,W] contiguous"""
    B, HW = x.shape[0], x.shape[1] * x.shape[2]
    if tuple(w_hwio.shape[-2:]) != (cin, C) or not w_hwio.is_contiguous() or w_hwio.dtype != torch.float32:
        raise capi.CutieError("head_fwd: weights must be contiguous f32 [1,1,%d,%d], got %s" % (cin, C, tuple(w_hwio.shape)))
    if probs is not None and (not probs.is_contiguous() or probs.numel() != B * HW * C):
        raise capi.CutieError("head_fwd: probs must be contiguous [B*H*W, C]")
    if labels is not None and (not labels.is_contiguous() or labels.numel() != B * HW):
        raise capi.CutieError("head_fwd: labels must be contiguous [B,H,W]")
    call("cutie_head_fwd", ptr(x), ld_of(x), dt_of(x), B, HW, cin, ptr(w_hwio), ptr(bias), C, eps, ptr(a), ld_of(a),
         ptr(stats), _tiles(stats, B, C, True), ptr(logits), ld_of(logits), ptr(probs), ptr(labels),
         _label_bytes(labels) if labels is not None else 4, ptr(ce), ptr(bad), ptr(workspace),
         workspace.numel() * workspace.element_size() if workspace is not None else 0, stream(x.device))


def head_bwd(a, stats, logits, probs, labels, ce, C, bias, ratio, weight, red, loss_out, dz, dbias_part, workspace,
             loss_scale=None, eps=IN_EPS):
    """hard-negative select + d(loss)/d(logits) + the logits layer's norm / ReLU gradient - cutie_head_bwd; must follow
    head_fwd (same workspace).  red f32 [B, head_chunks(HW), C, 2]; dz [B,H,W,>=C] view; dbias_part f32 [B*chunks, C]"""
    B, HW = a.shape[0], a.shape[1] * a.shape[2]
    cps = head_chunks(HW)
    if red.numel() != B * cps * C * 2 or not red.is_contiguous() or red.dtype != torch.float32:
        raise capi.CutieError("head_bwd: red must be contiguous f32 [B, %d, C, 2]" % cps)
    if dbias_part is not None and (dbias_part.numel() < B * cps * C or not dbias_part.is_contiguous()):
        raise capi.CutieError("head_bwd: dbias_part needs %d rows of C" % (B * cps))
    call("cutie_head_bwd", ptr(a), ld_of(a), ptr(stats), _tiles(stats, B, C, True), ptr(logits), ld_of(logits), ptr(probs),
         ptr(labels), _label_bytes(labels), ptr(ce), B, HW, C, eps, ptr(bias), int(ratio), ptr(loss_scale), ptr(weight),
         ptr(red), ptr(loss_out), ptr(dz), ld_of(dz), dt_of(dz), ptr(dbias_part), ptr(workspace),
         workspace.numel() * workspace.element_size(), stream(a.device))


def ghm_state(bins, device):
    """zeroed device state for ghm_weights (acc_sum carries over between steps)"""
    n = capi.lib().cutie_ghm_state_bytes(int(bins))
    if n == 0:
        raise capi.CutieError("ghm_bins=%r is outside 1..256" % (bins,))
    return torch.zeros((n + 7) // 8, dtype=torch.int64, device=device)


def ghm_weights(logits, labels, C, bins, momentum, eps, state, weights, weighted_logits=None):
    """logits [..., >=C] view; labels int32; weights / weighted_logits f32 [N, C] contiguous"""
    call("cutie_ghm_weights", ptr(logits), ld_of(logits), dt_of(logits), ptr(labels), labels.numel(), C, int(bins),
         float(momentum), float(eps), ptr(state), state.numel() * state.element_size(), ptr(weights),
         ptr(weighted_logits), stream())


# ------------------------------------------------------------------------------------ K12, K14
def l2_add(w, g, wd, reg_loss, scratch, loss_scale=None):
    """g = g / S + wd * w  (S = 1 without loss_scale);  reg_loss += wd * sum(w^2) / 2"""
    call("cutie_l2_add", ptr(w), ptr(g), w.numel(), float(wd), ptr(reg_loss), ptr(scratch), ptr(loss_scale), stream())


def sumsq(x, out, scratch, loss_scale=None):
    call("cutie_sumsq", ptr(x), x.numel(), dt_of(x), ptr(out), ptr(scratch), ptr(loss_scale), stream())


def loss_scale_state(scale, device):
    """device array {S, 1/S, good steps in a row, steps skipped, of them invalid, ...} (include/cutie_hip.h, K12)"""
    state = torch.zeros(capi.LOSS_SCALE_STATE, dtype=torch.float32, device=device)
    call("cutie_loss_scale_init", ptr(state), float(scale), stream())
    return state


def loss_scale_update(state, gnorm_sq, backoff=0.5, growth=2.0, interval=1000, min_scale=1.0, max_scale=65536.0,
                      invalid=None):
    call("cutie_loss_scale_update", ptr(state), ptr(gnorm_sq), float(backoff), float(growth), int(interval),
         float(min_scale), float(max_scale), ptr(invalid), stream())


def adam_step(p, g, m, v, lr_t, beta1, beta2, eps, gnorm_sq, clip, grad_scale=1.0, t=0, loss_scale=None, invalid=None):
    """t, loss_scale: the caller's step count behind lr_t and the loss-scale state - with both, the kernel re-bases the
    bias correction to the steps the device did not skip (an f16 overflow skips a step without the host knowing).
    invalid: int32 [1] device counter; non-zero skips the step (conv_dgrad_in_bwd's `late`)."""
    call("cutie_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr_t), float(beta1), float(beta2),
         float(eps), ptr(gnorm_sq), float(clip), float(grad_scale), int(t), ptr(loss_scale), ptr(invalid), stream())


def opt_layers(entries):
    """entries: (w_hwio view INTO the flat parameter buffer, w_fwd or None, w_dgrad or None) per conv layer, in the order
    of the flat buffer -> the host descriptor array cutie_opt_step walks (kept by the caller)"""
    if len(entries) > capi.MAX_BATCH:
        raise capi.CutieError("opt_layers: %d layers (max %d)" % (len(entries), capi.MAX_BATCH))
    arr = (capi.PackDesc * max(1, len(entries)))()
    for i, (w, wf, wd) in enumerate(entries):
        kh, kw, cin, cout = w.shape
        if not w.is_contiguous() or w.dtype != torch.float32:
            raise capi.CutieError("opt_layers: weights must be contiguous f32 HWIO views")
        arr[i].w_hwio, arr[i].w_fwd, arr[i].w_dgrad = ptr(w), ptr(wf), ptr(wd)
        arr[i].taps, arr[i].cin, arr[i].cout = kh * kw, cin, cout
        arr[i].cin_pad = wf.shape[2] if wf is not None else 0
        arr[i].cout_pad = wd.shape[2] if wd is not None else 0
    return arr, len(entries)


def opt_norm(w, g, n_l2, n_dense, wd, partials, rows=None, loss_scale=None):
    """pass 1 of the optimizer tail (cutie_opt_norm): partial sums of (g/S + wd*w)^2 and w^2; rows = the embedding's
    per-cell gradient rows (still scaled), contiguous"""
    if partials.numel() < capi.OPT_SCRATCH or partials.dtype != torch.float32:
        raise capi.CutieError("opt_norm: partials needs %d f32 elements" % capi.OPT_SCRATCH)
    call("cutie_opt_norm", ptr(w), ptr(g), int(n_l2), int(n_dense), float(wd), ptr(rows),
         rows.numel() if rows is not None else 0, dt_of(rows) if rows is not None else F32, ptr(loss_scale), ptr(partials),
         stream(w.device))


def opt_step(p, g, m, v, n_l2, wd, reg_scale, layers, dt, lr_t, beta1, beta2, eps, clip, grad_scale, t, partials,
             extra_sq=None, loss_scale=None, invalid=None, gnorm_sq_out=None, reg_out=None, zero_from=-1):
    """pass 2 (cutie_opt_step): clip + Adam over the flat buffers, the conv kernels in tiles that also write the packed
    16-bit operands; layers = opt_layers(...)"""
    arr, n = layers
    call("cutie_opt_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), int(n_l2), float(wd), float(reg_scale), arr, n, dt,
         float(lr_t), float(beta1), float(beta2), float(eps), float(clip), float(grad_scale), int(t), ptr(partials),
         ptr(extra_sq), ptr(loss_scale), ptr(invalid), ptr(gnorm_sq_out), ptr(reg_out), int(zero_from), stream(p.device))


def batch_scatter_add(dout, idx, dfeat):
    """dout [B, n_cells, >=C] view; idx int32 [B, n_cells]; dfeat f32 [B, n_src, C] contiguous, caller-zeroed."""
    B, n_src, C = dfeat.shape
    if dfeat.dtype != torch.float32 or not dfeat.is_contiguous():
        raise capi.CutieError("batch_scatter_add: dfeat must be contiguous f32")
    call("cutie_batch_scatter_add", ptr(dout), ld_of(dout), dt_of(dout), ptr(idx), idx.shape[1], n_src, B, C,
         ptr(dfeat), stream())
    return dfeat


def batch_gather(feat, idx, out, oob=None):
    """feat [B, n_src, C]; idx int32 [B, n_cells]; out [B, n_cells, >=C] view; oob int32 [1] counter of indices
    outside [0, n_src) (they are clamped; tf.batch_gather raises)."""
    B, n_src, C = feat.shape
    call("cutie_batch_gather", ptr(feat), ld_of(feat), n_src, ptr(idx), idx.shape[1], B, C, ptr(out), ld_of(out),
         dt_of(feat), ptr(oob), stream())
    return out


def convert(src, dst):
    C = src.shape[-1]
    n_rows = src.numel() // C
    call("cutie_convert", ptr(src), ld_of(src), dt_of(src), ptr(dst), ld_of(dst), dt_of(dst), n_rows, C, stream())
    return dst


def conv_clock_probe(probe):
    """Measurement hook (cutie_conv_clock_probe): probe = int64 device tensor of 3 zeroed words, or None to switch it off.
    While set, workgroup 0 of every wide conv launch adds {shader-clock cycles, 100 MHz ticks, 1} of its K loop to it."""
    if probe is not None and (probe.dtype != torch.int64 or probe.numel() < 3 or not probe.is_contiguous()):
        raise CutieError("conv_clock_probe wants a contiguous int64 device tensor of 3 words")
    call("cutie_conv_clock_probe", ptr(probe))
