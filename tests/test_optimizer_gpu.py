"""The optimizer tail in two launches (cutie_opt_norm / cutie_opt_step, main_train_json.py:152-156) against the separate
kernels it replaces (cutie_l2_add, cutie_sumsq, cutie_adam_step, cutie_pack_weights), and the device-side gate that skips
a step whose conv launches counted a tile that gave up waiting for its sample's other tiles."""
import math
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# the library's A/B switches (read once at load) that take the in-launch tile meeting away on purpose: nothing to test then
_NO_MEETING = any(os.environ.get(k) == "0" for k in ("CUTIE_RENDEZVOUS", "CUTIE_FUSE_NORM_BWD", "CUTIE_CONV_WIDE"))

DEV = "cuda:0"
V, E, C, H, W, B = 300, 128, 18, 32, 32, 2


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    grid = torch.zeros((B, H, W), dtype=torch.int32)
    occ = torch.rand((B, H, W), generator=g) < 0.15
    grid[occ] = torch.randint(1, V, (int(occ.sum()),), generator=g, dtype=torch.int32)
    labels = torch.zeros((B, H, W), dtype=torch.int32)
    fg = occ & (torch.rand((B, H, W), generator=g) < 0.3)
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g, dtype=torch.int32)
    return grid, labels


def _engine(dtype="f32", **kw):
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIERes(V, C, p)
    net.build_loss()
    return Engine(net, DEV, dtype=dtype, seed=0, **kw)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("clip", [10.0, 1e-3])
def test_apply_update_matches_the_separate_kernels(dtype, clip):
    """one step from the same raw gradients by both routes: parameters, moments, losses and norm agree to rounding (the
    two routes add the squared norm in different orders), and the packed operands are exactly the pack of the new
    parameters"""
    from cutie_amd import ops
    eng = _engine(dtype)
    grid, labels = _inputs(5)
    g = torch.Generator().manual_seed(3)
    m0 = (torch.randn(eng.n_total, generator=g) * 1e-3).to(DEV)
    v0 = (torch.rand(eng.n_total, generator=g) * 1e-6).to(DEV)
    used = torch.zeros(eng.n_total, dtype=torch.bool, device=DEV)
    for (off, n, _) in eng.offsets.values():
        used[off:off + n] = True
    m0[~used] = 0.0           # the padding between the tensors never sees a gradient: its moments are zero in any real run
    v0[~used] = 0.0
    P = eng.forward(grid, labels=labels)
    eng.loss_backward(P, defer_l2=True)
    raw = eng.grads_flat.clone()
    p0 = eng.params_flat.clone()
    lr_t, b1, b2, eps, t = 1e-3, 0.9, 0.999, 1e-8, 3
    # ---- the separate kernels, on copies
    ls = eng.loss_scale
    wd = 5e-4
    g_ref, p_ref, m_ref, v_ref = raw.clone(), p0.clone(), m0.clone(), v0.clone()
    reg = torch.zeros(1, dtype=torch.float32, device=DEV)
    scratch = torch.zeros(1024, dtype=torch.float32, device=DEV)
    ops.l2_add(p_ref[:eng.n_conv_w], g_ref[:eng.n_conv_w], wd, reg, scratch, ls)
    if ls is not None:
        ops.l2_add(p_ref[eng.n_conv_w:], g_ref[eng.n_conv_w:], 0.0, reg, scratch, ls)
    gn = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.sumsq(g_ref[:eng.n_dense], gn, scratch)
    ops.sumsq(P.d_embed.reshape(-1), gn, scratch, ls)
    ops.adam_step(p_ref, g_ref, m_ref, v_ref, lr_t, b1, b2, eps, gn, clip, t=t, loss_scale=eng.opt_state)
    # ---- the two-launch tail
    m, v = m0.clone(), v0.clone()
    gn2 = eng.apply_update(P, m, v, lr_t, b1, b2, eps, clip, world=1, t=t)
    torch.cuda.synchronize()
    assert abs(float(gn2) - float(gn)) <= 2e-6 * float(gn)
    assert abs(float(P.reg) - float(reg)) <= 2e-6 * float(reg)
    mv, mv_ref = (eng.params_flat - p0).double(), (p_ref - p0).double()
    assert float(mv_ref.abs().max()) > 0
    assert float((mv - mv_ref).norm() / mv_ref.norm()) < 1e-5
    assert float((m - m_ref).abs().max()) <= 1e-6 * float(m_ref.abs().max())
    assert float((v - v_ref).abs().max()) <= 1e-5 * float(v_ref.abs().max())
    # padding between the tensors stays untouched (zero parameters, zero moments as they were)
    assert torch.equal(eng.params_flat[~used], p0[~used]) and torch.equal(m[~used], m0[~used])
    # the embedding gradient is cleared behind the step, the rest of the buffer is as backward left it
    assert float(eng.grads_flat[eng.n_dense:].abs().max()) == 0.0
    assert torch.equal(eng.grads_flat[:eng.n_dense], raw[:eng.n_dense])
    # packed operands = pack(new parameters), bit for bit
    for name in eng.conv_w_names:
        wf, wd_ = ops.pack_weights(eng.params[name].contiguous(), eng.dt)
        assert torch.equal(wf.view(torch.int16 if dtype != "f32" else torch.int32),
                           eng.w_fwd[name].view(torch.int16 if dtype != "f32" else torch.int32)), name
        assert torch.equal(wd_.view(torch.int16 if dtype != "f32" else torch.int32),
                           eng.w_dg[name].view(torch.int16 if dtype != "f32" else torch.int32)), name


def test_opt_step_flat_only_and_skip_paths():
    """no layer table (nothing to repack): plain Adam over a flat buffer; a non-finite norm and a raised `invalid` word
    leave p, m, v untouched but still clear g[zero_from:]; the loss-scale state counts both kinds of skipped step"""
    from cutie_amd import capi, ops
    g = torch.Generator().manual_seed(11)
    n = 70000
    p0 = torch.randn(n, generator=g).to(DEV)
    gr = (torch.randn(n, generator=g) * 0.1).to(DEV)
    part = torch.zeros(capi.OPT_SCRATCH, dtype=torch.float32, device=DEV)
    state = ops.loss_scale_state(8.0, DEV)
    assert state.cpu().tolist() == [8.0, 0.125, 0, 0, 0, 0, 0, 0]
    gn = torch.zeros(1, dtype=torch.float32, device=DEV)
    inv = torch.zeros(1, dtype=torch.int32, device=DEV)
    layers = ops.opt_layers([])

    def run(grads, invalid):
        p, m, v, gg = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0), grads.clone()
        ops.opt_norm(p, gg, 0, n - 1000, 0.0, part, loss_scale=state)
        ops.opt_step(p, gg, m, v, 0, 0.0, 0.0, layers, capi.F32, 1e-2, 0.9, 0.999, 1e-8, 10.0, 1.0, 1, part,
                     loss_scale=state, invalid=invalid, gnorm_sq_out=gn, zero_from=n - 1000)
        ops.loss_scale_update(state, gn, interval=1000, invalid=invalid)
        torch.cuda.synchronize()
        return p, m, v, gg

    p, m, v, gg = run(gr, inv)
    want = float(((gr[:n - 1000].double() / 8.0) ** 2).sum())
    assert abs(float(gn) - want) < 1e-5 * want
    # first Adam step: m = (1 - b1) g', v = (1 - b2) g'^2, move = lr_t * m / (sqrt(v) + eps)
    gp = gr.double() / 8.0
    mv = 1e-2 * (0.1 * gp) / ((0.001 * gp * gp).sqrt() + 1e-8)
    assert float(((p0 - p).double() - mv).abs().max()) < 1e-5
    assert float(gg[n - 1000:].abs().max()) == 0.0 and torch.equal(gg[:n - 1000], gr[:n - 1000])
    assert state.cpu().tolist()[:5] == [8.0, 0.125, 1.0, 0.0, 0.0]
    bad = gr.clone()
    bad[17] = float("inf")
    p, m, v, gg = run(bad, inv)
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0
    assert float(gg[n - 1000:].abs().max()) == 0.0
    assert state.cpu().tolist()[:5] == [4.0, 0.25, 0.0, 1.0, 0.0]
    inv.fill_(3)
    p, m, v, gg = run(gr, inv)
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0
    assert float(gg[n - 1000:].abs().max()) == 0.0
    assert state.cpu().tolist()[:5] == [4.0, 0.25, 0.0, 2.0, 1.0]      # counted, the scale left alone


@pytest.mark.parametrize("where", ["forward", "backward"])
def test_a_late_tile_skips_the_step_on_the_device(where):
    """A conv tile whose bounded wait for its sample's other tiles runs out stores results from incomplete sums.  Forced
    here by poisoning one tile's tags in a layer's sync area (that tile then publishes a tag nobody else expects, and
    vice versa: every wave of the sample runs into the 0.2 s bound).  With Trainer() DEFAULTS - no host check - the
    parameters, moments and packed operands must come out of the step bit-identical, the device must have counted the
    step as invalid, and check_inputs() must say so afterwards; the step after that trains normally."""
    from cutie_amd.capi import CutieError
    from cutie_amd.train import Trainer
    eng = _engine("f16")
    tr = Trainer(eng, learning_rate=1e-3)
    assert tr.check_inputs is False
    grid, labels = _inputs(9)
    batch = {"grid_table": grid, "gt_classes": labels}
    out = tr.step(batch)
    assert np.isfinite(out["total_loss"]) and eng.steps_invalid() == 0
    P = eng._plans[(B, H, W, 0)]
    areas = P.sync_fw if where == "forward" else P.sync
    used = [k for k, t in areas.items() if int(t[16:].abs().max()) > 0]      # the areas a one-launch form has written tags into
    if not used and _NO_MEETING:
        pytest.skip("the tile meeting is switched off in the environment")
    assert used, "no layer took the one-launch %s form at this shape" % where
    area = areas[used[0]]
    # granules of (sample 0, tile 1): [B][T][n_out] pairs of 8-byte {value, tag}, 64 bytes in; the tag is the high word of
    # each half (2 * n_out 8-byte words a tile, all of them bumped)
    n_out = eng.net.layers[used[0]].channels
    g64 = area[16:].view(torch.int64).view(B, -1, 2, n_out)
    g64[0, 1] += (1000 << 32)
    before = (eng.params_flat.clone(), tr.m.clone(), tr.v.clone(),
              {k: t.clone() for k, t in eng.w_fwd.items()}, eng.opt_state.clone())
    out = tr.step(batch)
    torch.cuda.synchronize()
    assert torch.equal(eng.params_flat, before[0]) and torch.equal(tr.m, before[1]) and torch.equal(tr.v, before[2])
    for k, t in eng.w_fwd.items():
        assert torch.equal(t.view(torch.int16), before[3][k].view(torch.int16)), k
    st = eng.opt_state.cpu().tolist()
    assert st[4] == 1.0 and st[3] == before[4].cpu().tolist()[3] + 1.0 and st[0] == float(before[4][0])
    assert eng.steps_invalid() == 1
    with pytest.raises(CutieError, match="gave up waiting"):
        eng.check_inputs(P)
    # check_inputs has put the sync areas back to zero: the next step is a normal one
    out = tr.step(batch)
    torch.cuda.synchronize()
    assert np.isfinite(out["total_loss"]) and eng.steps_invalid() == 1
    assert not torch.equal(eng.params_flat, before[0])
    eng.check_inputs(P)
    # and with check_inputs=True the Trainer raises by itself, after the device has skipped the step
    tr2 = Trainer(eng, learning_rate=1e-3, check_inputs=True)
    g64[0, 1] += (1000 << 32)
    p_before = eng.params_flat.clone()
    with pytest.raises(CutieError, match="gave up waiting"):
        tr2.step(batch)
    assert torch.equal(eng.params_flat, p_before)


def test_sync_area_reused_with_another_geometry_is_cleared_by_the_library():
    """one buffer handed to cutie_conv_in_fwd for two different layers: the library notices the new geometry and starts
    the area from zero instead of letting the tiles derive tags from the other layer's granules (no late waves)"""
    from cutie_amd import ops
    g = torch.Generator().manual_seed(2)
    Bq, Hq, Wq = 2, 32, 32
    late = torch.zeros(1, dtype=torch.int32, device=DEV)
    T = ops.conv_stats_tiles(Hq, Wq)
    sync = torch.zeros(ops.conv_sync_bytes(Bq, T, 256) // 4, dtype=torch.int32, device=DEV)
    fused_any = False
    for cout in (256, 128, 256):
        x = torch.randn((Bq, Hq, Wq, 64), generator=g).to(DEV).half()
        w = (torch.randn((3, 5, 64, cout), generator=g) * 0.05).to(DEV)
        wf, _ = ops.pack_weights(w, 2)
        bias = torch.zeros(cout, device=DEV)
        a = torch.zeros((Bq, Hq, Wq, cout), dtype=torch.float16, device=DEV)
        y = torch.zeros_like(a)
        stats = torch.zeros((Bq, T, cout, 2), dtype=torch.float32, device=DEV)
        seg = [dict(x=x, w=wf, cin=64, kh=3, kw=5, dil=1)]
        for _ in range(2):
            fused_any |= ops.conv_in_fwd(seg, a, y, cout, bias, "relu", stats, sync=sync, late=late)
        torch.cuda.synchronize()
        assert int(late) == 0 and int(sync[0]) == 0
        yy = y.float()
        assert abs(float(yy.mean())) < 1e-2 and abs(float(yy.var(dim=(1, 2), unbiased=False).mean()) - 1.0) < 2e-2
    if not fused_any and _NO_MEETING:
        pytest.skip("the tile meeting is switched off in the environment")
    assert fused_any
