"""GPU parity tests, kernel by kernel, through the C ABI (cutie_amd.ops -> libcutie_hip.so).

Checker: oracle/ (CPU restatements; grid oracle pinned to the reference's golden vectors,
model ops on PyTorch CPU float64) and tests/golden fixtures.  Bars: integer / index work
bit-exact; f32 path 1e-4 relative to the tensor's max magnitude; bf16 path 2e-2 (inputs are
rounded to bf16 before both sides see them, so only accumulation/rounding differs).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# the library's A/B switches (read once at load, include/cutie_hip.h) that take the one-launch forms away on purpose
_TWO_CALL_BY_SWITCH = any(os.environ.get(k) == "0" for k in ("CUTIE_RENDEZVOUS", "CUTIE_FUSE_NORM_BWD", "CUTIE_CONV_WIDE"))

from oracle import cutie_oracle as co  # noqa: E402
from oracle import grid_oracle as go  # noqa: E402

DEV = "cuda:0"
TDT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
CDT = {"f32": 0, "bf16": 1, "f16": 2}     # include/cutie_hip.h CUTIE_F32 / CUTIE_BF16 / CUTIE_F16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cutie_amd import ops as _ops
    return _ops


def relerr(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def pack_docs(docs):
    box, ids, ecid, ext, off = [], [], [], [], [0]
    return box, ids, ecid, ext, off


def run_grid(ops, docs, rows, cols, strategy, num_classes):
    box, ids, ecid, ext, off = [], [], [], [], [0]
    for d in docs:
        for t in d:
            box.append(t["box"])
            ids.append(t["dict_id"])
            ecid.append(t["entity_id"] * num_classes + t["class_id"])
        ext.append(d[0]["extent"] if d else [0, 0, 1, 1])
        off.append(len(ids))
    ti = lambda a, shape: torch.tensor(a, dtype=torch.int32).reshape(shape).to(DEV)
    out = ops.grid_build(ti(box, (-1, 4)), ti(ids, (-1,)), ti(ecid, (-1,)), ti(ext, (-1, 4)), ti(off, (-1,)),
                         rows, cols, strategy)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, off


# ------------------------------------------------------------------------------------ K0
def test_grid_build_faktura_golden(ops, golden_dir):
    with open(os.path.join(golden_dir, "faktura1_items.json"), encoding="utf-8") as f:
        items = json.load(f)
    with open(os.path.join(golden_dir, "faktura1_grids.json")) as f:
        meta = json.load(f)
    npz = np.load(os.path.join(golden_dir, "faktura1_grids.npz"))
    doc = items["docs"]["training"][0]
    nc = items["num_classes"]
    n = 0
    for key, m in meta.items():
        if "error" in m:
            continue
        out, off = run_grid(ops, [doc], m["rows"], m["cols"], m["strategy"], nc)
        assert out["status"].tolist() == [0], key
        np.testing.assert_array_equal(out["grid"], npz[key + "_grid"][..., 0], err_msg=key)
        np.testing.assert_array_equal(out["labels"], npz[key + "_labels"], err_msg=key)
        if m["strategy"] == 1:
            np.testing.assert_array_equal(out["ps_x"], npz[key + "_ps_x"], err_msg=key)
            np.testing.assert_array_equal(out["ps_y"], npz[key + "_ps_y"], err_msg=key)
        ref = go.positional_mapping([doc], m["rows"], m["cols"], m["strategy"], nc)
        assert out["order"].tolist() == ref["order"][0], key
        assert [tuple(p) for p in out["place"].tolist()] == ref["placed"][0], key
        n += 1
    assert n >= 12


def test_grid_build_synthetic_golden(ops, golden_dir):
    with open(os.path.join(golden_dir, "synthetic_grids.json")) as f:
        meta = json.load(f)
    npz = np.load(os.path.join(golden_dir, "synthetic_grids.npz"))
    nc = meta["num_classes"]
    for case in meta["cases"]:
        name = case["name"]
        out, off = run_grid(ops, case["docs"], case["rows"], case["cols"], case["strategy"], nc)
        assert not out["status"].any(), name
        np.testing.assert_array_equal(out["grid"], npz[name + "_grid"][..., 0], err_msg=name)
        np.testing.assert_array_equal(out["labels"], npz[name + "_labels"], err_msg=name)
        if case["strategy"] == 1:
            np.testing.assert_array_equal(out["ps_x"], npz[name + "_ps_x"], err_msg=name)
            np.testing.assert_array_equal(out["ps_y"], npz[name + "_ps_y"], err_msg=name)
        ref = go.positional_mapping(case["docs"], case["rows"], case["cols"], case["strategy"], nc)
        for d in range(len(case["docs"])):
            assert out["order"][off[d]:off[d + 1]].tolist() == ref["order"][d], name
            assert [tuple(p) for p in out["place"][off[d]:off[d + 1]].tolist()] == ref["placed"][d], name


def test_grid_build_edges(ops):
    # empty batch entry, out-of-grid token -> status 1, large grid 256x128
    tok = {"box": [10, 10, 30, 20], "extent": [8, 8, 100, 100], "dict_id": 7, "class_id": 2, "entity_id": 0}
    bad = {"box": [0, 0, 10, 10], "extent": [0, 0, 5, 5], "dict_id": 3, "class_id": 0, "entity_id": 0}
    out, _ = run_grid(ops, [[tok], [bad], [tok, tok]], 256, 128, 1, 18)
    assert out["status"].tolist() == [0, 1, 0]
    ref = go.positional_mapping([[tok], [tok, tok]], 256, 128, 1, 18)
    np.testing.assert_array_equal(out["grid"][0], ref["grid"][0])
    np.testing.assert_array_equal(out["grid"][2], ref["grid"][1])
    np.testing.assert_array_equal(out["labels"][2], ref["labels"][1])


@pytest.mark.parametrize("strategy", [0, 1])
def test_grid_build_long_document(ops, strategy):
    """2600 tokens in one document: more than the LDS token scratch holds (2048), so the builder keeps its keys in
    the workspace and ranks by comparing all pairs; next to it a short document on the row-grouped path."""
    rng = np.random.RandomState(5)
    ext = [0, 0, 1500, 2100]

    def doc(n):
        toks = []
        for _ in range(n):
            x, y = int(rng.randint(0, 1400)), int(rng.randint(0, 2050))
            toks.append({"box": [x, y, x + int(rng.randint(5, 90)), y + int(rng.randint(8, 40))], "extent": ext,
                         "dict_id": int(rng.randint(0, 50)), "class_id": int(rng.randint(0, 18)), "entity_id": 0})
        return toks
    docs = [doc(2600), doc(300)]
    out, off = run_grid(ops, docs, 64, 64, strategy, 18)
    ref = go.positional_mapping(docs, 64, 64, strategy, 18)
    assert out["status"].tolist() == [0, 0]
    np.testing.assert_array_equal(out["grid"], ref["grid"])
    np.testing.assert_array_equal(out["labels"], ref["labels"])
    for d in range(2):
        assert out["order"][off[d]:off[d + 1]].tolist() == ref["order"][d]
        assert [tuple(p) for p in out["place"][off[d]:off[d + 1]].tolist()] == ref["placed"][d]


# ------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_embed_fwd_bwd(ops, dt):
    g = torch.Generator().manual_seed(1)
    V, E, n = 300, 128, 5000
    tdt = TDT[dt]
    table = torch.rand((V, E), generator=g) * 2 - 1
    ids = torch.randint(0, V, (n,), generator=g, dtype=torch.int32)
    ids[torch.rand(n, generator=g) < 0.6] = 0
    ids[torch.rand(n, generator=g) < 0.1] = 1
    ids[17] = V + 5      # out of range -> zero row + counter
    ids[18] = -1
    out = torch.full((n, E + 64), 7.0, dtype=tdt, device=DEV)
    oob = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embed_fwd(ids.to(DEV), table.to(DEV), out[:, :E], oob)
    ref = table[ids.clamp(0, V - 1).long()].clone()
    ref[17] = 0
    ref[18] = 0
    assert int(oob.item()) == 2
    got = out[:, :E].float().cpu()
    if dt == "f32":
        assert torch.equal(got, ref)
    else:
        assert torch.equal(got, ref.to(tdt).float())
    assert torch.all(out[:, E:] == 7.0)
    # backward: scatter-add
    dout = (torch.randn((n, E), generator=g)).to(tdt)
    dtab = torch.zeros((V, E), dtype=torch.float32, device=DEV)
    ops.embed_bwd(ids.to(DEV), dout.to(DEV), dtab)
    valid = (ids >= 0) & (ids < V)
    refg = torch.zeros((V, E), dtype=torch.float64)
    refg.index_add_(0, ids[valid].long(), dout[valid].double())
    assert relerr(dtab, refg) < 1e-5
    # with the workspace: rows 0 and 1 folded in a fixed order - the same bits every run
    ws = torch.empty(ops.embed_bwd_workspace_bytes(n, E), dtype=torch.uint8, device=DEV)
    got = []
    for _ in range(2):
        dtab.zero_()
        ops.embed_bwd(ids.to(DEV), dout.to(DEV), dtab, ws)
        got.append(dtab.clone())
    assert relerr(got[0], refg) < 1e-5
    assert torch.equal(got[0][:2], got[1][:2])
    # more distinct ids per workgroup than its LDS table holds (the overflow adds straight into the table),
    # and one id that every cell of a workgroup shares
    for ids2 in (torch.arange(n, dtype=torch.int32) % (V - 2) + 2,
                 torch.where(torch.arange(n) % 3 == 0, torch.full((n,), 7), torch.arange(n) % 5).to(torch.int32)):
        dtab.zero_()
        ops.embed_bwd(ids2.to(DEV), dout.to(DEV), dtab, ws)
        refg = torch.zeros((V, E), dtype=torch.float64)
        refg.index_add_(0, ids2.long(), dout.double())
        assert relerr(dtab, refg) < 1e-5


# ------------------------------------------------------------------------------------ K2-K5
def _rand_act(g, shape, tdt):
    return (torch.randn(shape, generator=g) * 0.7).to(tdt)


CONV_CASES = [
    # B, H, W, cin, cout, kh, kw, dil
    (2, 20, 64, 64, 128, 3, 5, 1),
    (1, 40, 64, 128, 96, 3, 5, 2),
    (1, 20, 64, 64, 128, 3, 5, 4),
    (1, 24, 64, 64, 64, 3, 5, 8),
    (1, 40, 64, 64, 256, 3, 5, 16),
    (2, 9, 40, 64, 18, 1, 1, 1),
    (1, 7, 80, 64, 160, 3, 5, 2),
    (1, 5, 130, 64, 64, 3, 5, 1),
    (1, 12, 12, 64, 64, 3, 5, 1),
    (1, 6, 128, 64, 32, 3, 3, 2),
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(ops, dt, case):
    from cutie_amd import capi
    B, H, W, cin, cout, kh, kw, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    tdt = TDT[dt]
    cdt = CDT[dt]
    x = _rand_act(g, (B, H, W, cin), tdt)
    w = (torch.randn((kh, kw, cin, cout), generator=g) * 0.05).to(tdt).float()
    bias = torch.randn(cout, generator=g) * 0.1
    ref_z = co.conv_raw(x.double(), w.double(), dil) + bias.double()
    ref_a = torch.relu(ref_z)
    wf, _ = ops.pack_weights(w.to(DEV), cdt)
    ld_out = cout + 8
    out = torch.full((B, H, W, ld_out), 3.0, dtype=torch.float32, device=DEV)
    stats = torch.zeros((B, ops.conv_stats_tiles(H, W) + 1, cout, 2), dtype=torch.float32, device=DEV)   # per-tile slots
    xd = torch.zeros((B, H, W, cin + 64), dtype=tdt, device=DEV)   # read through a channel-slice view
    xd[..., 32:32 + cin] = x.to(DEV)
    ops.conv_fwd([dict(x=xd[..., 32:32 + cin], w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out[..., :cout], cout,
                 bias=bias.to(DEV), relu=True, stats=stats)
    torch.cuda.synchronize()
    tol = 1e-4 if dt == "f32" else 2e-3      # f32 accumulate of bf16 products; inputs identical
    assert relerr(out[..., :cout], ref_a) < tol
    assert torch.all(out[..., cout:] == 3.0)
    sh = torch.relu(bias.double())                          # the moments are kept about act(bias) (cutie_hip.h, K6)
    ref_s = torch.stack([(ref_a - sh).sum(dim=(1, 2)), ((ref_a - sh) ** 2).sum(dim=(1, 2))], dim=-1)
    assert relerr(stats.sum(dim=1), ref_s) < (1e-4 if dt == "f32" else 2e-3)
    assert float(stats[:, -1].abs().max()) == 0.0           # a slot no tile owns stays untouched
    # no float atomics: a second launch reproduces output and moments bit for bit
    out2, stats2 = torch.full_like(out, 3.0), torch.zeros_like(stats)
    ops.conv_fwd([dict(x=xd[..., 32:32 + cin], w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out2[..., :cout], cout,
                 bias=bias.to(DEV), relu=True, stats=stats2)
    assert torch.equal(out, out2) and torch.equal(stats, stats2)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_conv_bf16_out_and_no_relu(ops, dt):
    from cutie_amd import capi
    B, H, W, cin, cout, kh, kw, dil = 1, 16, 64, 64, 128, 3, 5, 2
    g = torch.Generator().manual_seed(5)
    tdt = TDT[dt]
    cdt = CDT[dt]
    x = _rand_act(g, (B, H, W, cin), tdt)
    w = (torch.randn((kh, kw, cin, cout), generator=g) * 0.05).to(tdt).float()
    ref = co.conv_raw(x.double(), w.double(), dil)
    wf, _ = ops.pack_weights(w.to(DEV), cdt)
    out = torch.zeros((B, H, W, cout), dtype=torch.float16 if dt == "f16" else torch.bfloat16, device=DEV)
    ops.conv_fwd([dict(x=x.to(DEV), w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out, cout)
    assert relerr(out, ref) < (8e-4 if dt == "f16" else 6e-3)     # one 16-bit rounding of the output


WIDE_CASES = [
    # B, H, W, cin, cout, kh, kw, dil  -- bf16 in / bf16 out with cout % 256 == 0: the LDS-DMA kernel
    (2, 64, 64, 64, 256, 3, 5, 1),
    (1, 64, 64, 128, 256, 3, 5, 16),
    (1, 32, 64, 64, 256, 3, 5, 8),
    (1, 42, 63, 64, 256, 3, 5, 4),
    (1, 7, 80, 192, 512, 3, 5, 2),
    (3, 16, 16, 320, 256, 1, 1, 1),
    (1, 5, 130, 64, 256, 3, 3, 2),
    # channel tiles 128 and 64 (decoder 512 -> 64, the data gradient into the 128-channel embedding)
    (2, 64, 64, 256, 128, 3, 5, 1),
    (1, 64, 64, 512, 64, 3, 5, 1),
    (1, 33, 50, 64, 384, 3, 5, 4),
    (2, 20, 64, 128, 192, 1, 1, 1),
    (1, 40, 64, 64, 64, 3, 5, 16),
    # B % 8 == 0 with unequal live taps per tile: the heaviest-first tile order
    (8, 64, 64, 64, 256, 3, 5, 16),
    (8, 32, 64, 64, 128, 3, 5, 8),
    (16, 20, 40, 64, 64, 3, 3, 4),
    # fewer than 64 output channels: one 64-channel tile with masked rows / stores (CUTIE2's image branch)
    (2, 40, 64, 64, 32, 3, 3, 2),
    (1, 33, 50, 64, 48, 3, 5, 1),
    (8, 16, 64, 64, 8, 3, 3, 1),
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("case", WIDE_CASES)
def test_conv_wide_16bit(ops, case, dt):
    B, H, W, cin, cout, kh, kw, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 7)
    tdt = TDT[dt]
    x = _rand_act(g, (B, H, W, cin), tdt)
    w = (torch.randn((kh, kw, cin, cout), generator=g) * 0.05).to(tdt).float()
    bias = torch.randn(cout, generator=g) * 0.1
    ref_a = torch.relu(co.conv_raw(x.double(), w.double(), dil) + bias.double())
    wf, _ = ops.pack_weights(w.to(DEV), CDT[dt])
    ld_out = cout + 8
    out = torch.full((B, H, W, ld_out), 3.0, dtype=tdt, device=DEV)
    stats = torch.zeros((B, ops.conv_stats_tiles(H, W) + 1, cout, 2), dtype=torch.float32, device=DEV)   # per-tile slots
    xd = torch.zeros((B, H, W, cin + 64), dtype=tdt, device=DEV)
    xd[..., 32:32 + cin] = x.to(DEV)
    ops.conv_fwd([dict(x=xd[..., 32:32 + cin], w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out[..., :cout], cout,
                 bias=bias.to(DEV), relu=True, stats=stats)
    torch.cuda.synchronize()
    assert relerr(out[..., :cout], ref_a) < (8e-4 if dt == "f16" else 6e-3)            # one rounding of the output
    # against the rounded reference the only differences are last-bit rounding flips
    assert (out[..., :cout].float().cpu() - ref_a.float().to(tdt).float()).abs().max() < (0.009 if dt == "f16" else 0.07)
    assert torch.all(out[..., cout:] == 3.0)
    sh = torch.relu(bias.double())                          # the moments are kept about act(bias)
    ref_s = torch.stack([(ref_a - sh).sum(dim=(1, 2)), ((ref_a - sh) ** 2).sum(dim=(1, 2))], dim=-1)
    assert relerr(stats.sum(dim=1), ref_s) < 2e-3
    out2, stats2 = torch.full_like(out, 3.0), torch.zeros_like(stats)
    ops.conv_fwd([dict(x=xd[..., 32:32 + cin], w=wf, cin=cin, kh=kh, kw=kw, dil=dil)], out2[..., :cout], cout,
                 bias=bias.to(DEV), relu=True, stats=stats2)
    assert torch.equal(out, out2) and torch.equal(stats, stats2)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_conv_wide_dgrad_multi_segment(ops, dt):
    """data gradient of a 256-channel tensor consumed by a 3x5 dil 4 conv (256 outputs), a 3x5
    dil 16 conv (256 outputs) and channels [256:512) of a 1x1 conv over a 768-channel concat."""
    tdt = TDT[dt]
    g = torch.Generator().manual_seed(19)
    B, H, W, C = 1, 32, 64, 256
    w1 = (torch.randn((3, 5, C, 256), generator=g) * 0.03).to(tdt).float()
    w2 = (torch.randn((3, 5, C, 256), generator=g) * 0.03).to(tdt).float()
    w3 = (torch.randn((1, 1, 768, 128), generator=g) * 0.03).to(tdt).float()
    dz1 = _rand_act(g, (B, H, W, 256), tdt)
    dz2 = _rand_act(g, (B, H, W, 256), tdt)
    dz3 = _rand_act(g, (B, H, W, 128), tdt)
    x = torch.zeros((B, H, W, C), dtype=torch.float64, requires_grad=True)
    z = torch.zeros((B, H, W, 256), dtype=torch.float64)
    y1 = co.conv_raw(x, w1.double(), 4)
    y2 = co.conv_raw(x, w2.double(), 16)
    y3 = co.conv_raw(torch.cat([z, x, z], dim=3), w3.double(), 1)
    ((y1 * dz1.double()).sum() + (y2 * dz2.double()).sum() + (y3 * dz3.double()).sum()).backward()
    wd = [ops.pack_weights(w.to(DEV), CDT[dt], want_fwd=False)[1] for w in (w1, w2, w3)]
    out = torch.zeros((B, H, W, C), dtype=tdt, device=DEV)
    ops.conv_fwd([dict(x=dz1.to(DEV), w=wd[0], cin=256, kh=3, kw=5, dil=4, flip=True),
                  dict(x=dz2.to(DEV), w=wd[1], cin=256, kh=3, kw=5, dil=16, flip=True),
                  dict(x=dz3.to(DEV), w=wd[2], cin=128, kh=1, kw=1, dil=1, flip=True, n_off=256)], out, C)
    assert relerr(out, x.grad) < (8e-4 if dt == "f16" else 6e-3)


DGRAD_STATS_CASES = [
    # B, H, W, channels of the consuming conv's output (cin of the data gradient), channels of dy, dil
    (2, 64, 64, 256, 256, 4),      # wide kernel, channel tile 256: sums taken in the epilogue
    (1, 42, 63, 64, 256, 1),       # ragged tiles
    (2, 32, 64, 64, 128, 2),       # channel tile 128 (two workgroups per pixel tile)
    (1, 40, 64, 128, 64, 16),      # channel tile 64
    (1, 7, 80, 64, 512, 2),        # two channel tiles of 256, tiles wider than tall
    (2, 12, 64, 64, 32, 1),        # fewer than 64 channels: masked 64-channel tile
    (2, 12, 64, 64, 36, 1),        # not a wide-kernel shape: conv + the separate reduction behind the same entry point
    (8, 64, 64, 64, 256, 16),      # heaviest-first tile order: slots are still indexed by the tile, not by the block
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", DGRAD_STATS_CASES)
def test_conv_dgrad_in_stats(ops, dt, case):
    """cutie_conv_dgrad_in_stats = cutie_conv_fwd (flip) followed by cutie_in_bwd_stats: the same dy bit for bit,
    the same (sum dy, sum dy*xhat) up to the order of the f32 additions; rerun bit-identical."""
    B, H, W, cz, C, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 3)
    tdt = TDT[dt]
    w = (torch.randn((3, 5, C, cz), generator=g) * 0.05).to(tdt).float()     # forward weights C -> cz
    dz = _rand_act(g, (B, H, W, cz), tdt).to(DEV)
    a = torch.relu(torch.randn((B, H, W, C), generator=g) + 0.2).to(tdt)
    ld = (C + 15) // 8 * 8
    ad = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
    ad[..., :C] = a.to(DEV)
    T = ops.conv_stats_tiles(H, W)
    stats = torch.zeros((B, 3, C, 2), dtype=torch.float32, device=DEV)
    ops.in_stats(ad[..., :C], stats)
    _, wd = ops.pack_weights(w.to(DEV), CDT[dt], want_fwd=False)
    seg = [dict(x=dz, w=wd, cin=wd.shape[2], kh=3, kw=5, dil=dil, flip=True)]
    if wd.shape[2] != cz:
        pytest.skip("padded cin")
    dy_ref = torch.full((B, H, W, ld), 2.0, dtype=tdt, device=DEV)
    ops.conv_fwd(seg, dy_ref[..., :C], C)
    red_ref = torch.zeros((B, 4, C, 2), dtype=torch.float32, device=DEV)
    ops.in_bwd_stats(dy_ref[..., :C], ad[..., :C], stats, red_ref)
    dy = torch.full((B, H, W, ld), 2.0, dtype=tdt, device=DEV)
    red = torch.zeros((B, T + 2, C, 2), dtype=torch.float32, device=DEV)
    ops.conv_fwd(seg, dy[..., :C], C, norm_bwd=(ad[..., :C], stats, red))
    torch.cuda.synchronize()
    assert torch.equal(dy, dy_ref)
    # the sums cancel heavily (dy has mean ~0): compare against the magnitude of the summands
    scale = torch.stack([dy_ref[..., :C].float().abs().sum(dim=(1, 2)), dy_ref[..., :C].float().abs().sum(dim=(1, 2))], dim=-1)
    err = ((red.sum(dim=1) - red_ref.sum(dim=1)).abs() / (scale + 1e-20)).max()
    assert float(err) < 1e-5
    # against float64 of the same rounded operands
    a64, d64 = ad[..., :C].double().cpu(), dy_ref[..., :C].double().cpu()
    mean = a64.mean(dim=(1, 2), keepdim=True)
    var = (a64 ** 2).mean(dim=(1, 2), keepdim=True) - mean ** 2
    xh = (a64 - mean) / torch.sqrt(var.clamp_min(0) + 1e-6)
    ref64 = torch.stack([d64.sum(dim=(1, 2)), (d64 * xh).sum(dim=(1, 2))], dim=-1)
    err64 = ((red.sum(dim=1).double().cpu() - ref64).abs() / (scale.double().cpu() * 2 + 1e-20)).max()
    assert float(err64) < 1e-4
    dy2, red2 = torch.full_like(dy, 2.0), torch.zeros_like(red)
    ops.conv_fwd(seg, dy2[..., :C], C, norm_bwd=(ad[..., :C], stats, red2))
    assert torch.equal(dy, dy2) and torch.equal(red, red2)


CONV_IN_FWD_CASES = [
    # B, H, W, cin, cout, kh, kw, dil
    (32, 64, 64, 64, 256, 3, 5, 1),      # the bench shape: 16 tiles a sample, two rounds of workgroups
    (5, 64, 64, 64, 256, 3, 5, 4),       # grid not a multiple of 8: a sample's tiles spread over the XCDs
    (3, 42, 63, 64, 128, 3, 5, 2),       # ragged tiles, channel tile 128
    (4, 40, 64, 128, 64, 1, 1, 1),       # 1x1, channel tile 64
    (2, 7, 80, 64, 512, 3, 5, 2),        # two channel tiles of 256
    (8, 64, 64, 64, 256, 3, 5, 16),      # unbalanced tiles: the entry point takes the two-call form itself
    (2, 12, 64, 64, 36, 3, 5, 1),        # not a wide-kernel shape: two-call form
    (1, 160, 128, 64, 128, 3, 5, 1),     # 80 tiles a sample: more granules to sweep than the pass is worth - two-call form
    (8, 64, 128, 64, 512, 3, 5, 1),      # 32 tiles x 2 channel tiles a sample: plain block order
]


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("case", CONV_IN_FWD_CASES)
def test_conv_in_fwd_one_launch_matches_the_two_call_form(ops, dt, case):
    """cutie_conv_in_fwd with a sync area (one launch: a sample's tiles meet, y is stored, a is not) against the same
    entry point without one (conv into a, then cutie_in_apply): the moments bit for bit, y to a unit in the last place
    (rounded once from the f32 activation instead of from the stored 16-bit one) and against float64; a sample's y does
    not depend on the batch it sits in.  Five launches on fresh data: a stale granule would show."""
    B, H, W, cin, cout, kh, kw, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 11)
    tdt = TDT[dt]
    w = (torch.randn((kh, kw, cin, cout), generator=g) * 0.05).to(tdt).float()
    bias = torch.randn(cout, generator=g) * 0.1
    wf, _ = ops.pack_weights(w.to(DEV), CDT[dt], want_dgrad=False)
    seg = lambda x: [dict(x=x, w=wf, cin=wf.shape[2], kh=kh, kw=kw, dil=dil)]
    if wf.shape[2] != cin:
        pytest.skip("padded cin")
    T = ops.conv_stats_tiles(H, W)
    ld = cout + 8
    sync = torch.zeros(ops.conv_sync_bytes(B, T, cout) // 4, dtype=torch.int32, device=DEV)
    bd = bias.to(DEV)
    for it in range(5):
        x = _rand_act(g, (B, H, W, cin), tdt).to(DEV)
        a_ref = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        y_ref = torch.full((B, H, W, ld), 3.0, dtype=tdt, device=DEV)
        st_ref = torch.zeros((B, T, cout, 2), dtype=torch.float32, device=DEV)
        ops.conv_in_fwd(seg(x), a_ref[..., :cout], y_ref[..., :cout], cout, bd, True, st_ref, sync=None)
        a = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        y = torch.full((B, H, W, ld), 3.0, dtype=tdt, device=DEV)
        st = torch.zeros((B, T, cout, 2), dtype=torch.float32, device=DEV)
        ops.conv_in_fwd(seg(x), a[..., :cout], y[..., :cout], cout, bd, True, st, sync=sync)
        torch.cuda.synchronize()
        assert int(sync[0]) == 0, "a tile's wait ran out"
        assert torch.equal(st, st_ref), it
        assert torch.all(y[..., cout:] == 3.0)
        d = (y[..., :cout].float() - y_ref[..., :cout].float()).abs()
        ulp = {"f16": 2.0 ** -10, "bf16": 2.0 ** -7, "f32": 1e-6}[dt]
        assert float((d / (y_ref[..., :cout].float().abs() + 1.0)).max()) <= 2.02 * ulp, it      # + the ReLU-pattern bump
    fused = int((sync[16:].view(torch.int64) >> 32).abs().sum()) > 0
    if fused:
        assert dt != "f32" and float(a.float().abs().max()) == 0.0       # the pre-norm activation was never written
        # one sample alone gives the same bits as inside the batch
        y1 = torch.zeros((1, H, W, ld), dtype=tdt, device=DEV)
        s1 = torch.zeros(ops.conv_sync_bytes(1, T, cout) // 4, dtype=torch.int32, device=DEV)
        ops.conv_in_fwd(seg(x[B - 1:]), torch.zeros_like(y1)[..., :cout], y1[..., :cout], cout, bd, True,
                        torch.zeros((1, T, cout, 2), dtype=torch.float32, device=DEV), sync=s1)
        assert torch.equal(y1[..., :cout], y[B - 1:, :, :, :cout])
    else:
        assert _TWO_CALL_BY_SWITCH or dt == "f32" or dil == 16 or cout == 36 or H * W > 32 * 256
    z = co.conv_raw(x.double().cpu(), w.double(), dil) + bias.double()
    ref = co.instance_norm(torch.relu(z))
    assert relerr(y[..., :cout], ref) < {"f32": 1e-4, "f16": 3e-3, "bf16": 2e-2}[dt]


@pytest.mark.parametrize("dt", ["f16", "bf16"])
@pytest.mark.parametrize("case", CONV_IN_FWD_CASES[:5])
def test_backward_from_y_alone(ops, dt, case):
    """The layer gradient from the one-launch forward's output ALONE (CUTIE_ACT_FROM_Y: xhat = y, the ReLU pattern read
    out of y's bits) against the same gradient from the stored pre-norm activation: the pattern element for element
    (up to f32 activations that underflow in 16 bits), dz within the 16-bit tolerance and against float64; the
    one-launch backward and the two-call backward agree bit for bit on the same operands."""
    B, H, W, cin, cout, kh, kw, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 23)
    tdt = TDT[dt]
    w = (torch.randn((kh, kw, cin, cout), generator=g) * 0.05).to(tdt).float()
    bias = torch.randn(cout, generator=g) * 0.1
    bias[0] = 12.0                                      # a channel its bias dominates: no zeros in its planes
    bias[1] = -40.0                                     # and a dead one: every element is y0
    wf, _ = ops.pack_weights(w.to(DEV), CDT[dt], want_dgrad=False)
    if wf.shape[2] != cin:
        pytest.skip("padded cin")
    cz = 64
    w2 = (torch.randn((3, 5, cout, cz), generator=g) * 0.05).to(tdt).float()
    _, wd = ops.pack_weights(w2.to(DEV), CDT[dt], want_fwd=False)
    T = ops.conv_stats_tiles(H, W)
    bd = bias.to(DEV)
    x = _rand_act(g, (B, H, W, cin), tdt).to(DEV)
    seg = [dict(x=x, w=wf, cin=cin, kh=kh, kw=kw, dil=dil)]
    a = torch.zeros((B, H, W, cout), dtype=tdt, device=DEV)
    y2 = torch.zeros((B, H, W, cout), dtype=tdt, device=DEV)
    st2 = torch.zeros((B, T, cout, 2), dtype=torch.float32, device=DEV)
    assert not ops.conv_in_fwd(seg, a, y2, cout, bd, True, st2, sync=None)
    y = torch.zeros((B, H, W, cout), dtype=tdt, device=DEV)
    st = torch.zeros((B, T, cout, 2), dtype=torch.float32, device=DEV)
    sync_fw = torch.zeros(ops.conv_sync_bytes(B, T, cout) // 4, dtype=torch.int32, device=DEV)
    if not ops.conv_in_fwd(seg, torch.zeros_like(a), y, cout, bd, True, st, sync=sync_fw):
        pytest.skip("the entry point took the two-call form")
    dzin = _rand_act(g, (B, H, W, cz), tdt).to(DEV)
    bseg = [dict(x=dzin, w=wd, cin=wd.shape[2], kh=3, kw=5, dil=1, flip=True)]
    out = {}
    for key, av, from_y, use_sync in (("a", a, False, True), ("y", y, True, True), ("y2", y, True, False)):
        dy = torch.zeros((B, H, W, cout), dtype=tdt, device=DEV)
        dz = torch.zeros((B, H, W, cout), dtype=tdt, device=DEV)
        red = torch.zeros((B, T, cout, 2), dtype=torch.float32, device=DEV)
        dbp = torch.zeros((B * T, cout), dtype=torch.float32, device=DEV)
        sync = torch.zeros(ops.conv_sync_bytes(B, T, cout) // 4, dtype=torch.int32, device=DEV) if use_sync else None
        ops.conv_dgrad_in_bwd(bseg, dy, cout, av, st, red, dz, dbias_part=dbp, relu=True, sync=sync, bias=bd, from_y=from_y)
        torch.cuda.synchronize()
        assert sync is None or int(sync[0]) == 0
        out[key] = (dz, dbp.view(B, T, cout).sum(dim=1))
    assert torch.equal(out["y"][0], out["y2"][0])                         # one launch == two calls, from y
    active = a > 0
    got = out["y"][0] != 0
    wrong = int((got & ~active).sum()), int((~got & active & (out["a"][0] != 0)).sum())
    print("pattern read from y: %d set where a == 0, %d clear where a > 0, of %d" % (wrong + (active.numel(),)))
    assert max(wrong) <= max(4, active.numel() // 1000000)
    assert bool(active[..., 0].all()) and not bool(active[..., 1].any())
    assert not bool(got[..., 1].any())
    tol = {"f16": 2e-3, "bf16": 1.2e-2}[dt]
    # (channel 0 aside: there the stored 16-bit activation is the coarse operand, y the fine one)
    # and the few elements aside whose f32 activation is positive but underflows in 16 bits: y knows they are active
    same = (got == (out["a"][0] != 0))
    assert relerr((out["y"][0] * same)[..., 2:], (out["a"][0] * same)[..., 2:]) < tol
    scale = out["a"][0].float().abs().sum(dim=(1, 2))
    assert float(((out["y"][1] - out["a"][1]).abs() / (scale + 1e-20))[:, 2:].max()) < tol


DGRAD_BWD_CASES = [
    # B, H, W, channels of the consuming conv's output, channels of dy, dil, activation
    (32, 64, 64, 64, 256, 1, "relu"),    # the bench shape: 16 tiles a sample, a sample's tiles on one XCD, two rounds of workgroups
    (5, 64, 64, 64, 256, 2, "relu"),     # grid not a multiple of 8: a sample's tiles spread over the XCDs
    (3, 42, 63, 64, 256, 1, "relu"),     # ragged tiles
    (8, 32, 64, 64, 128, 4, "none"),     # channel tile 128; no activation
    (4, 40, 64, 128, 64, 1, "relu"),     # channel tile 64
    (2, 7, 80, 64, 512, 2, "relu"),      # two channel tiles of 256: two counters per sample
    (8, 64, 64, 64, 256, 16, "relu"),    # unbalanced tiles (dilation 16) with a short K loop: either form
    (32, 64, 64, 256, 256, 16, "relu"),  # the same with the model's K loop: the entry point takes the two-call form itself
    (2, 12, 64, 64, 36, 1, "relu"),      # not a wide-kernel shape: two-call form
    (1, 160, 128, 64, 128, 1, "relu"),   # 80 tiles a sample: the sweep costs more than the pass - two-call form
    (8, 64, 128, 64, 512, 1, "relu"),    # 32 tiles x 2 channel tiles a sample: plain block order
]


@pytest.mark.parametrize("dt", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("case", DGRAD_BWD_CASES)
def test_conv_dgrad_in_bwd_matches_the_two_call_form(ops, dt, case):
    """cutie_conv_dgrad_in_bwd with a sync area (one launch, the sample's tiles meet inside it) against the same entry
    point without one (data gradient, then cutie_in_bwd_apply): dz bit for bit - same operands, same order - and the
    bias-gradient partials up to the order of the f32 additions.  Five launches in a row on fresh data each: a slot or
    a granule read stale from a cache would show as a mismatch; the tags must have advanced by one per launch."""
    B, H, W, cz, C, dil, act = case
    g = torch.Generator().manual_seed(hash(case[:6]) % 1000 + 5)
    tdt = TDT[dt]
    w = (torch.randn((3, 5, C, cz), generator=g) * 0.05).to(tdt).float()
    _, wd = ops.pack_weights(w.to(DEV), CDT[dt], want_fwd=False)
    if wd.shape[2] != cz:
        pytest.skip("padded cin")
    ld = (C + 15) // 8 * 8
    T = ops.conv_stats_tiles(H, W)
    sync = torch.zeros(ops.conv_sync_bytes(B, T, C) // 4, dtype=torch.int32, device=DEV)
    red = torch.zeros((B, T, C, 2), dtype=torch.float32, device=DEV)
    dbp = torch.zeros((B * T, C), dtype=torch.float32, device=DEV)
    stats = torch.zeros((B, 3, C, 2), dtype=torch.float32, device=DEV)
    for it in range(5):
        dzin = _rand_act(g, (B, H, W, cz), tdt).to(DEV)
        a = torch.relu(torch.randn((B, H, W, C), generator=g) + 0.2 * it).to(tdt)
        ad = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        ad[..., :C] = a.to(DEV)
        ops.in_stats(ad[..., :C], stats)
        seg = [dict(x=dzin, w=wd, cin=wd.shape[2], kh=3, kw=5, dil=dil, flip=True)]
        dy_ref = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        dz_ref = torch.full((B, H, W, ld), 3.0, dtype=tdt, device=DEV)
        red_ref = torch.zeros((B, T, C, 2), dtype=torch.float32, device=DEV)
        dbp_ref = torch.zeros((B * T, C), dtype=torch.float32, device=DEV)
        ops.conv_dgrad_in_bwd(seg, dy_ref[..., :C], C, ad[..., :C], stats, red_ref, dz_ref[..., :C], dbias_part=dbp_ref,
                              relu=act, sync=None)
        dy = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        dz = torch.full((B, H, W, ld), 3.0, dtype=tdt, device=DEV)
        ops.conv_dgrad_in_bwd(seg, dy[..., :C], C, ad[..., :C], stats, red, dz[..., :C], dbias_part=dbp, relu=act, sync=sync)
        torch.cuda.synchronize()
        assert int(sync[0]) == 0, "a tile's wait ran out"
        assert torch.equal(dz.view(torch.int16) if dt != "f32" else dz, dz_ref.view(torch.int16) if dt != "f32" else dz_ref), it
        db, db_ref = dbp.view(B, T, C).sum(dim=1), dbp_ref.view(B, T, C).sum(dim=1)
        scale = dz_ref[..., :C].float().abs().sum(dim=(1, 2))
        assert float(((db - db_ref).abs() / (scale + 1e-20)).max()) < 1e-5, it
    tags = (sync[16:].view(torch.int64) >> 32).cpu()
    if int(tags.abs().sum()) > 0:                      # the one-launch form ran: every granule it wrote carries tag 5
        assert bool(((tags == 5) | (tags == 0)).all()) and int((tags == 5).sum()) >= 2 * B * C
        assert dt != "f32" and float(dy.float().abs().max()) == 0.0      # and dy was never written
    else:
        assert _TWO_CALL_BY_SWITCH or dt == "f32" or case[5] == 16 or case[4] == 36 or case[1] * case[2] > 32 * 256      # only those may take the two-call form
    # against float64 of the same rounded operands (last iteration)
    a64, d64 = ad[..., :C].double().cpu(), dy_ref[..., :C].double().cpu()
    mean = a64.mean(dim=(1, 2), keepdim=True)
    var = ((a64 - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    inv = 1.0 / torch.sqrt(var + 1e-6)
    xh = (a64 - mean) * inv
    ref = inv * (d64 - d64.mean(dim=(1, 2), keepdim=True) - xh * (d64 * xh).mean(dim=(1, 2), keepdim=True))
    if act == "relu":
        ref = ref * (a64 > 0)
    assert relerr(dz[..., :C], ref) < {"f32": 1e-4, "f16": 2e-3, "bf16": 1.2e-2}[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_conv_dgrad_multi_segment(ops, dt):
    """d(loss)/dx of a tensor consumed by two convs (3x5 dil 2 with 128 outputs, and channels
    [0:64) of a 3x5 dil 1 conv over a 128-channel concat with 64 outputs) in ONE launch."""
    from cutie_amd import capi
    g = torch.Generator().manual_seed(9)
    tdt = TDT[dt]
    cdt = CDT[dt]
    B, H, W, C = 2, 12, 64, 64
    w1 = (torch.randn((3, 5, C, 128), generator=g) * 0.05).to(tdt).float()
    w2 = (torch.randn((3, 5, 128, 64), generator=g) * 0.05).to(tdt).float()
    dz1 = _rand_act(g, (B, H, W, 128), tdt)
    dz2 = _rand_act(g, (B, H, W, 64), tdt)
    x = torch.zeros((B, H, W, C), dtype=torch.float64, requires_grad=True)
    other = torch.zeros((B, H, W, 64), dtype=torch.float64)
    y1 = co.conv_raw(x, w1.double(), 2)
    y2 = co.conv_raw(torch.cat([x, other], dim=3), w2.double(), 1)
    ((y1 * dz1.double()).sum() + (y2 * dz2.double()).sum()).backward()
    _, wd1 = ops.pack_weights(w1.to(DEV), cdt, want_fwd=False)
    _, wd2 = ops.pack_weights(w2.to(DEV), cdt, want_fwd=False)
    out = torch.zeros((B, H, W, C), dtype=torch.float32, device=DEV)
    ops.conv_fwd([dict(x=dz1.to(DEV), w=wd1, cin=128, kh=3, kw=5, dil=2, flip=True),
                  dict(x=dz2.to(DEV), w=wd2, cin=64, kh=3, kw=5, dil=1, flip=True, n_off=0)], out, C)
    assert relerr(out, x.grad) < (1e-4 if dt == "f32" else 2e-3)


WGRAD_CASES = [
    (2, 10, 64, 64, 128, 3, 5, 1),
    (1, 40, 64, 128, 64, 3, 5, 16),
    (3, 6, 40, 64, 18, 1, 1, 1),
    (1, 9, 80, 96, 72, 3, 5, 2),
    (1, 5, 130, 64, 64, 3, 3, 4),
    (8, 64, 64, 128, 128, 3, 5, 16),    # many rows per workgroup: both LDS halves, dead (K block, tap) pairs
    (8, 64, 64, 128, 128, 3, 5, 8),
    # several 64-pixel segments per row with short shifts: the all-live compile-time pattern (also on a ragged last segment)
    (1, 6, 128, 64, 128, 3, 5, 8),
    (2, 5, 100, 64, 128, 3, 5, 2),
    (1, 4, 70, 64, 128, 3, 5, 1),
    # dilation 16 on an image wider than one 64-pixel segment: 128-row patch (64 + 2 x 32 halo)
    (2, 8, 128, 64, 128, 3, 5, 16),
    (1, 6, 100, 128, 128, 3, 5, 16),
    # 1x1 layers: two cin tiles per workgroup (ragged second tile, several tile pairs, one tile only)
    (2, 16, 64, 320, 128, 1, 1, 1),
    (1, 8, 64, 1024, 256, 1, 1, 1),
    (2, 9, 48, 160, 256, 1, 1, 1),
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(ops, dt, case):
    B, H, W, cin, cout, kh, kw, dil = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + 1)
    tdt = TDT[dt]
    x = _rand_act(g, (B, H, W, cin), tdt)
    dz = _rand_act(g, (B, H, W, cout), tdt)
    w = torch.zeros((kh, kw, cin, cout), dtype=torch.float64, requires_grad=True)
    (co.conv_raw(x.double(), w, dil) * dz.double()).sum().backward()
    ldz = (cout + 7) // 8 * 8
    dzd = torch.zeros((B, H, W, ldz), dtype=tdt, device=DEV)
    dzd[..., :cout] = dz.to(DEV)
    dw = torch.full((kh, kw, cin, cout), 9.0, dtype=torch.float32, device=DEV)
    ws = torch.empty(ops.conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw), dtype=torch.uint8, device=DEV)
    ops.conv_wgrad(x.to(DEV), cin, dzd[..., :cout], cout, kh, kw, dil, dw, ws)
    assert relerr(dw, w.grad) < (1e-4 if dt == "f32" else 2e-3)
    ops.conv_wgrad(x.to(DEV), cin, dzd[..., :cout], cout, kh, kw, dil, dw, ws, accumulate=True)
    assert relerr(dw, 2 * w.grad) < (1e-4 if dt == "f32" else 2e-3)


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_conv_wgrad_batched_matches_layer_by_layer(ops, dt):
    """cutie_conv_wgrad_batched over a mixed bag of layers - three on one kernel instance (one launch, fewer K splits), a
    dilation-16 layer and a 1x1 layer on instances of their own, a narrow one, one that accumulates - against
    cutie_conv_wgrad layer by layer: the same sums up to the order of the f32 slab additions, and against float64.
    f32 is not a batched shape: the call then runs the layers one by one and must give the same bits."""
    g = torch.Generator().manual_seed(17)
    tdt = TDT[dt]
    B, H, W = 4, 32, 64
    layers = [(128, 256, 3, 5, 1, False), (256, 128, 3, 5, 2, False), (128, 128, 3, 5, 4, True), (128, 128, 3, 5, 16, False),
              (256, 128, 1, 1, 1, False), (128, 64, 3, 5, 1, False), (64, 18, 1, 1, 1, False)]
    tasks, singles, refs = [], [], []
    for cin, cout, kh, kw, dil, acc in layers:
        x = _rand_act(g, (B, H, W, cin), tdt).to(DEV)
        ld = (cout + 7) // 8 * 8
        dz = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
        dz[..., :cout] = _rand_act(g, (B, H, W, cout), tdt).to(DEV)
        base = torch.randn((kh, kw, cin, cout), generator=g).to(DEV) if acc else torch.full((kh, kw, cin, cout), 7.0, device=DEV)
        dw_b, dw_s = base.clone(), base.clone()
        tasks.append((x, cin, dz[..., :cout], cout, kh, kw, dil, dw_b, acc))
        ws = torch.empty(ops.conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw), dtype=torch.uint8, device=DEV)
        ops.conv_wgrad(x, cin, dz[..., :cout], cout, kh, kw, dil, dw_s, ws, accumulate=acc)
        singles.append((dw_s, base if acc else None))
    arr, dtc = ops.wgrad_descs(tasks)
    need = ops.conv_wgrad_batched_workspace_bytes(arr, dtc)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    for rep in range(2):                     # a second call into the same workspace (the accumulating layer adds again)
        ops.conv_wgrad_batched(arr, dtc, ws)
    torch.cuda.synchronize()
    for (x, cin, dz, cout, kh, kw, dil, dw_b, acc), (dw_s, base) in zip(tasks, singles):
        if acc:
            got, want = dw_b - base, 2 * (dw_s - base)
        else:
            got, want = dw_b, dw_s
        if dt == "f32":
            assert torch.equal(got, want) or acc, (cin, cout, dil)
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale, (cin, cout, kh, kw, dil)
    # one layer against float64
    x, cin, dz, cout, kh, kw, dil, dw_b, _ = tasks[1]
    w = torch.zeros((kh, kw, cin, cout), dtype=torch.float64, requires_grad=True)
    (co.conv_raw(x.double().cpu(), w, dil) * dz.double().cpu()).sum().backward()
    assert relerr(dw_b, w.grad) < (1e-4 if dt == "f32" else 2e-3)
    with pytest.raises(Exception, match="workspace"):
        ops.conv_wgrad_batched(arr, dtc, ws[:need // 2])


# ------------------------------------------------------------------------------------ K6, K8
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("C", [256, 64, 18])
def test_instance_norm_fwd_bwd(ops, dt, C):
    g = torch.Generator().manual_seed(3)
    tdt = TDT[dt]
    B, H, W = 2, 20, 24
    a = torch.relu(torch.randn((B, H, W, C), generator=g) + 0.3).to(tdt)
    a[0, :, :, 1] = 0     # all-zero channel -> y = 0
    dy = torch.randn((B, H, W, C), generator=g).to(tdt)
    a64 = a.double().requires_grad_(True)
    # relu is the identity on a >= 0 with mask [a > 0]
    pre = a64
    y_ref = co.instance_norm(pre)
    ld = (C + 3) // 4 * 4 + 4
    ad = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
    ad[..., :C] = a.to(DEV)
    stats = torch.full((B, 5, C, 2), 7.0, dtype=torch.float32, device=DEV)      # 5 slices of H*W, every slot written
    ops.in_stats(ad[..., :C], stats)
    ref_s = torch.stack([a.double().sum(dim=(1, 2)), (a.double() ** 2).sum(dim=(1, 2))], dim=-1)
    assert relerr(stats.sum(dim=1), ref_s) < 1e-5
    y = torch.full((B, H, W, ld), 5.0, dtype=torch.float32, device=DEV)
    ops.in_apply(ad[..., :C], stats, y[..., :C])
    assert relerr(y[..., :C], y_ref) < 1e-4
    assert torch.all(y[..., C:] == 5.0)
    assert float(y[0, :, :, 1].abs().max()) == 0.0
    # backward through IN and the ReLU mask
    (y_ref * dy.double()).sum().backward()
    da_ref = a64.grad * (a.double() > 0)
    red = torch.full((B, 3, C, 2), 7.0, dtype=torch.float32, device=DEV)
    ops.in_bwd_stats(dy.to(DEV), ad[..., :C], stats, red)
    dz = torch.zeros((B, H, W, ld), dtype=tdt, device=DEV)
    dbp = torch.full((B * 4, C), 7.0, dtype=torch.float32, device=DEV)
    db = torch.full((C,), 4.0, dtype=torch.float32, device=DEV)
    ops.in_bwd_apply(dy.to(DEV), ad[..., :C], stats, red, dz[..., :C], dbias_part=dbp, relu=True)
    ops.sum_rows(dbp, db)
    tol = 1e-4 if dt == "f32" else 8e-3
    assert relerr(dz[..., :C], da_ref) < tol
    assert relerr(db, da_ref.sum(dim=(0, 1, 2))) < tol


def test_instance_norm_channel_dominated_by_its_bias(ops):
    """A channel with mean >> std (bias 50, sigma 0.01: the amplification case of SURVEY.md 7 - the norm multiplies it
    by ~100): f32 E[x^2] - E[x]^2 loses the whole variance (2500.0001 - 2500), so the moments are kept about
    act(bias).  Forward and backward of the norm against float64, through the conv epilogue's moments and through
    cutie_in_stats, f32 path."""
    g = torch.Generator().manual_seed(8)
    B, H, W, Cin, C = 2, 40, 64, 64, 128
    x = torch.randn((B, H, W, Cin), generator=g)
    w = torch.randn((3, 5, Cin, C), generator=g) * 0.05
    bias = torch.zeros(C)
    w[..., 5] *= 0.01 / 0.05 / (15 * Cin) ** 0.5       # channel 5: conv output sigma ~0.01 ...
    bias[5] = 50.0                                      # ... on top of a bias of 50
    w[..., 9] *= 0.02
    bias[9] = -30.0                                     # channel 9: dead behind the ReLU (all zeros, shift 0)
    wf, _ = ops.pack_weights(w.to(DEV), CDT["f32"], want_dgrad=False)
    T = ops.conv_stats_tiles(H, W)
    a = torch.zeros((B, H, W, C), dtype=torch.float32, device=DEV)
    stats = torch.zeros((B, T, C, 2), dtype=torch.float32, device=DEV)
    bd = bias.to(DEV)
    ops.conv_fwd([dict(x=x.to(DEV), w=wf, cin=wf.shape[2], kh=3, kw=5, dil=1)], a, C, bias=bd, relu=True, stats=stats)
    y = torch.zeros_like(a)
    ops.in_apply(a, stats, y, bias=bd, relu=True)
    z64 = co.conv_raw(x.double(), w.double(), 1) + bias.double()
    a64 = torch.relu(z64).requires_grad_(True)
    y_ref = co.instance_norm(a64)
    std5 = float(a64[..., 5].std())
    assert 0.005 < std5 < 0.02 and float(a64[..., 5].mean()) > 49
    assert relerr(a, a64) < 1e-6
    e_all = relerr(y, y_ref)
    e5 = float((y[..., 5].double().cpu() - y_ref[..., 5]).abs().max() / y_ref[..., 5].abs().max())
    print("mean >> std channel: y error %.2e (all channels %.2e)" % (e5, e_all))
    assert e5 < 2e-3 and e_all < 2e-3                   # a itself carries 50 * 2^-24 = 3e-6 of rounding on sigma 0.01
    assert float(y[..., 9].abs().max()) == 0.0
    # the same moments from cutie_in_stats, and the norm's gradient
    stats2 = torch.zeros((B, 4, C, 2), dtype=torch.float32, device=DEV)
    ops.in_stats(a, stats2, bias=bd, relu=True)
    y2 = torch.zeros_like(a)
    ops.in_apply(a, stats2, y2, bias=bd, relu=True)
    assert relerr(y2, y) < 1e-3
    dy = torch.randn((B, H, W, C), generator=g)
    (y_ref * dy.double()).sum().backward()
    da_ref = a64.grad * (a64.detach() > 0)
    red = torch.zeros((B, 3, C, 2), dtype=torch.float32, device=DEV)
    ops.in_bwd_stats(dy.to(DEV), a, stats, red, bias=bd, relu=True)
    dz = torch.zeros_like(a)
    ops.in_bwd_apply(dy.to(DEV), a, stats, red, dz, relu=True, bias=bd)
    e5 = float((dz[..., 5].double().cpu() - da_ref[..., 5]).abs().max() / da_ref[..., 5].abs().max())
    print("mean >> std channel: dz error %.2e" % e5)
    assert e5 < 5e-3 and relerr(dz, da_ref) < 5e-3
    # without the shift the channel's variance is gone: this is what the shift is for
    stats0 = torch.zeros((B, 4, C, 2), dtype=torch.float32, device=DEV)
    ops.in_stats(a, stats0)
    y0 = torch.zeros_like(a)
    ops.in_apply(a, stats0, y0)
    e0 = float((y0[..., 5].double().cpu() - y_ref[..., 5]).abs().max() / y_ref[..., 5].abs().max())
    print("mean >> std channel without the shift: y error %.2e" % e0)
    assert e0 > 10 * e5


@pytest.mark.parametrize("with_scratch", [False, True])
@pytest.mark.parametrize("dt,C", [("f32", 256), ("bf16", 256), ("f16", 32), ("f32", 20)])
def test_spatial_mean_bcast(ops, dt, C, with_scratch):
    g = torch.Generator().manual_seed(4)
    tdt = TDT[dt]
    x = (torch.randn((3, 10, 12, C), generator=g) + 0.5).to(tdt)
    out = torch.zeros((3, 10, 12, 3 * C + C), dtype=tdt, device=DEV)
    scratch = torch.full((3 * 32 * C * 2 + 5,), 9.0, dtype=torch.float32, device=DEV) if with_scratch else None   # CUTIE_POOL_TILES
    ops.spatial_mean_bcast(x.to(DEV), out[..., 3 * C:], scratch)
    assert relerr(out[..., 3 * C:], co.global_pool(x.double())) < {"f32": 1e-5, "bf16": 4e-3, "f16": 5e-4}[dt]
    assert float(out[..., :3 * C].abs().max()) == 0.0
    if with_scratch:
        assert torch.all(scratch[-5:] == 9.0)


# ------------------------------------------------------------------------------------ K9-K11
def _loss_case(ops, logits, labels, ratio, C):
    n = labels.numel()
    ld = logits.shape[-1]
    ce = torch.zeros(n, dtype=torch.float32, device=DEV)
    probs = torch.zeros((n, C), dtype=torch.float32, device=DEV)
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    weight = torch.zeros(n, dtype=torch.float32, device=DEV)
    loss = torch.zeros(4, dtype=torch.float32, device=DEV)
    dl = torch.zeros((n, ld), dtype=torch.float32, device=DEV)
    ld_ = logits.to(DEV)
    lab = labels.to(torch.int32).to(DEV)
    ops.softmax_ce_fwd(ld_[:, :C], lab, C, ce, probs, bad)
    ops.hnm_weights(ce, lab, ratio, weight, loss)
    ops.softmax_ce_bwd(probs, lab, weight, C, dl[:, :C])
    torch.cuda.synchronize()
    return ce.cpu(), probs.cpu(), int(bad.item()), weight.cpu(), loss.cpu(), dl.cpu()


@pytest.mark.parametrize("C,n,fgfrac,ratio", [(18, 5000, 0.02, 3), (9, 777, 0.3, 3), (5, 4096, 0.001, 3),
                                              (18, 300, 0.5, 1), (18, 131072, 0.03, 3), (4, 70001, 0.2, 2)])
def test_softmax_ce_hnm(ops, C, n, fgfrac, ratio):
    g = torch.Generator().manual_seed(C * 31 + n)
    ld = (C + 3) // 4 * 4
    logits = torch.zeros((n, ld))
    logits[:, :C] = torch.randn((n, C), generator=g) * 2
    labels = torch.zeros(n, dtype=torch.int64)
    fg = torch.rand(n, generator=g) < fgfrac
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g)
    # force ties at the selection threshold: duplicate some background rows
    logits[100:140] = logits[100]
    labels[100:140] = 0
    if n > 20000:       # ties spread over many 1024-cell chunks, some of them at the threshold
        bgi = torch.nonzero(labels == 0).flatten()
        ce0 = co.softmax_ce(logits[:, :C].double(), labels)
        kth = int(min(len(bgi), ratio * int((labels != 0).sum())))
        thr_row = bgi[torch.argsort(ce0[bgi], descending=True)[kth - 1]]
        dup = bgi[torch.randperm(len(bgi), generator=g)[:300]]
        logits[dup] = logits[thr_row].clone()
    ce, probs, bad, weight, loss, dl = _loss_case(ops, logits, labels, ratio, C)
    l64 = logits[:, :C].double().requires_grad_(True)
    ce_ref = co.softmax_ce(l64, labels)
    loss_ref, w_ref, nfg, nbg = co.hnm_loss(ce_ref, labels, ratio)
    assert bad == 0
    assert relerr(ce, ce_ref) < 1e-5
    assert relerr(probs, torch.softmax(l64, -1)) < 1e-5
    assert int(loss[1]) == nfg and int(loss[2]) == nbg
    # the GPU ranks by its own f32 ce; compare selection against the oracle run on those values
    _, w_sel, _, _ = co.hnm_loss(ce.double(), labels, ratio)
    assert torch.equal(weight != 0, w_sel != 0)
    assert relerr(weight, w_sel) < 1e-6
    assert abs(float(loss[0]) - float(loss_ref.detach())) < 1e-4 * abs(float(loss_ref.detach()))
    (w_sel * ce_ref).sum().backward()
    assert relerr(dl[:, :C], l64.grad) < 1e-4
    assert float(dl[:, C:].abs().max()) == 0.0 if ld > C else True


def test_hnm_edges(ops):
    C = 6
    logits = torch.randn((64, 8))
    # no foreground -> NaN loss, zero weights
    ce, probs, bad, weight, loss, dl = _loss_case(ops, logits, torch.zeros(64, dtype=torch.int64), 3, C)
    assert np.isnan(float(loss[0])) and float(weight.abs().max()) == 0.0
    # all foreground -> Nbg = 0 -> NaN loss
    ce, probs, bad, weight, loss, dl = _loss_case(ops, logits, torch.ones(64, dtype=torch.int64), 3, C)
    assert np.isnan(float(loss[0]))
    # ratio*Nfg >= Nbg -> every background cell selected
    labels = torch.zeros(64, dtype=torch.int64)
    labels[:30] = 2
    ce, probs, bad, weight, loss, dl = _loss_case(ops, logits, labels, 3, C)
    assert int(loss[2]) == 34 and int((weight != 0).sum()) == 64
    # label >= C is reported, not silently used
    labels[5] = C + 3
    ce, probs, bad, weight, loss, dl = _loss_case(ops, logits, labels, 3, C)
    assert bad == 1 and np.isnan(float(ce[5]))


# ------------------------------------------------------------------------------------ K12, K14
def test_l2_norm_adam(ops):
    g = torch.Generator().manual_seed(11)
    n = 1_000_003
    p = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * 0.05
    m = torch.randn(n, generator=g) * 0.01
    v = torch.rand(n, generator=g) * 0.01
    pd, gd, md, vd = (t.clone().to(DEV) for t in (p, gr, m, v))
    scratch = torch.zeros(1024, dtype=torch.float32, device=DEV)
    reg = torch.zeros(1, dtype=torch.float32, device=DEV)
    wd = 5e-4
    ops.l2_add(pd, gd, wd, reg, scratch)
    g_ref = gr.double() + wd * p.double()
    assert relerr(gd, g_ref) < 1e-6
    assert abs(float(reg) - float(wd * 0.5 * (p.double() ** 2).sum())) < 1e-4 * float(reg)
    nsq = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.sumsq(gd, nsq, scratch)
    nsq2 = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.sumsq(gd, nsq2, scratch)
    assert float(nsq) == float(nsq2)            # fixed-order reduction: bit-identical reruns
    assert abs(float(nsq) - float((g_ref ** 2).sum())) < 1e-4 * float(nsq)
    for clip in (10.0, 1.0):                    # inactive and active clipping
        pd2, md2, vd2 = pd.clone(), md.clone(), vd.clone()
        lr, b1, b2, eps, t = 1e-4, 0.9, 0.999, 1e-8, 3
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        ops.adam_step(pd2, gd, md2, vd2, lr_t, b1, b2, eps, nsq, clip)
        state = {"t": t - 1, "m": {"w": m.double()}, "v": {"w": v.double()}}
        newp = co.adam_step_tf({"w": p.double()}, {"w": g_ref}, state, lr, torch.sqrt((g_ref ** 2).sum()), clip=clip)
        # the update (~1e-4) is read back as a difference of f32 parameters (~1): ulp/update ~ 6e-4
        assert relerr(pd2.double() - pd.double(), newp["w"] - p.double()) < 2e-3
        assert relerr(pd2, newp["w"]) < 1e-6
        assert relerr(md2, state["m"]["w"]) < 1e-5 and relerr(vd2, state["v"]["w"]) < 1e-5


def test_batch_gather_and_convert(ops):
    g = torch.Generator().manual_seed(12)
    feat = torch.randn((2, 50, 32), generator=g)
    idx = torch.randint(-3, 60, (2, 40), generator=g, dtype=torch.int32)
    out = torch.zeros((2, 40, 160), dtype=torch.float32, device=DEV)
    ops.batch_gather(feat.to(DEV), idx.to(DEV), out[..., :32])
    ref = torch.stack([feat[b, idx[b].clamp(0, 49).long()] for b in range(2)])
    assert torch.equal(out[..., :32].cpu(), ref)
    dst = torch.zeros((2, 50, 32), dtype=torch.bfloat16, device=DEV)
    ops.convert(feat.to(DEV), dst)
    assert torch.equal(dst.cpu(), feat.bfloat16())


# ------------------------------------------------------------------------------------ batched per-layer work
@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_batched_pack_and_sum_rows_match_single(ops, dt):
    """One launch over a descriptor table has to write what the per-layer entry points write."""
    dt = CDT[dt]
    g = torch.Generator().manual_seed(13)
    shapes = [(3, 5, 128, 256), (1, 1, 256, 18), (3, 5, 40, 24), (1, 1, 1280, 256)] * 9   # > MAX_BATCH layers
    ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    singles = [ops.pack_weights(w, dt) for w in ws]
    entries = []
    for i, (w, (wf, wd)) in enumerate(zip(ws, singles)):
        wf2 = torch.full_like(wf, 7) if i % 3 != 1 else None          # some layers want one layout only
        wd2 = torch.full_like(wd, 7) if i % 3 != 2 else None
        entries.append((w, wf2, wd2))
    batches = ops.pack_weights_batched(entries, dt)
    assert len(batches) == 2
    for (w, wf2, wd2), (wf, wd) in zip(entries, singles):
        assert wf2 is None or torch.equal(wf2, wf)
        assert wd2 is None or torch.equal(wd2, wd)
    ws[0].mul_(2.0)                                                   # replay reads the current weights
    ops.replay_pack(batches, dt, DEV)
    assert torch.equal(entries[0][1], ops.pack_weights(ws[0], dt)[0])

    B = 5
    parts = [torch.randn((B, c), generator=g).to(DEV) for c in (256, 18, 24, 1000, 3)]
    outs = [torch.randn(p.shape[1], generator=g).to(DEV) for p in parts]
    for acc in (False, True):
        ref = [ops.sum_rows(p, o.clone(), accumulate=acc) for p, o in zip(parts, outs)]
        got = [o.clone() for o in outs]
        ops.sum_rows_batched(list(zip(parts, got)), B, accumulate=acc)
        for r, o in zip(ref, got):
            assert torch.equal(r, o)
    from cutie_amd.capi import CutieError
    with pytest.raises(CutieError):
        ops.sum_rows_batched([(parts[0], outs[1])], B)


# ------------------------------------------------------------------------------------ loss scale (f16 training)
def test_loss_scale_state_machine(ops):
    """softmax_ce_bwd puts S on the gradients, l2_add / sumsq take it out, a non-finite norm skips
    the Adam step and halves S, a run of finite steps doubles it - all without a host round trip."""
    g = torch.Generator().manual_seed(21)
    ls = ops.loss_scale_state(1024.0, DEV)
    assert ls.cpu().tolist()[:4] == [1024.0, 1.0 / 1024.0, 0.0, 0.0]
    n, C = 1000, 18
    probs = torch.softmax(torch.randn((n, C), generator=g), dim=-1).to(DEV)
    labels = torch.randint(0, C, (n,), generator=g, dtype=torch.int32).to(DEV)
    weight = (torch.rand(n, generator=g) * (torch.rand(n, generator=g) < 0.5)).to(DEV) * 1e-3
    d1 = torch.zeros((n, C), dtype=torch.float32, device=DEV)
    dS = torch.zeros((n, C), dtype=torch.float16, device=DEV)
    ops.softmax_ce_bwd(probs, labels, weight, C, d1)
    ops.softmax_ce_bwd(probs, labels, weight, C, dS, ls)
    assert relerr(dS.float() / 1024.0, d1.double()) < 6e-4          # one f16 rounding; unscaled it would lose the small ones
    # l2_add: g/S + wd*w, sumsq: sum/S^2
    w = torch.randn(5000, generator=g).to(DEV)
    gr = (torch.randn(5000, generator=g) * 300).to(DEV)
    scratch = torch.zeros(1024, dtype=torch.float32, device=DEV)
    reg = torch.zeros(1, dtype=torch.float32, device=DEV)
    g2 = gr.clone()
    ops.l2_add(w, g2, 5e-4, reg, scratch, ls)
    assert relerr(g2, gr.double() / 1024.0 + 5e-4 * w.double()) < 1e-6
    assert abs(float(reg) - 2.5e-4 * float((w.double() ** 2).sum())) < 1e-5 * float(reg)
    nsq = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.sumsq(gr.half(), nsq, scratch, ls)
    assert abs(float(nsq) - float((gr.half().double() ** 2).sum()) / 1024.0 ** 2) < 1e-5 * float(nsq)
    # Adam: finite norm -> update; non-finite -> nothing moves
    p, m, v = w.clone(), torch.zeros_like(w), torch.zeros_like(w)
    ops.adam_step(p, g2, m, v, 1e-3, 0.9, 0.999, 1e-8, nsq, 10.0)
    assert not torch.equal(p, w)
    for bad in (float("inf"), float("nan")):
        p2, m2, v2 = p.clone(), m.clone(), v.clone()
        nbad = torch.full((1,), bad, dtype=torch.float32, device=DEV)
        ops.adam_step(p2, g2, m2, v2, 1e-3, 0.9, 0.999, 1e-8, nbad, 10.0)
        assert torch.equal(p2, p) and torch.equal(m2, m) and torch.equal(v2, v)
    # bias correction after skipped steps: the caller's t = 5 with 3 steps skipped is Adam's step 2
    import math
    c = lambda t: math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    skipped = torch.tensor([1024.0, 1 / 1024.0, 0.0, 3.0], device=DEV)
    pa, ma, va = w.clone(), m.clone(), v.clone()
    pb, mb, vb = w.clone(), m.clone(), v.clone()
    ops.adam_step(pa, g2, ma, va, 1e-3 * c(5), 0.9, 0.999, 1e-8, nsq, 10.0, t=5, loss_scale=skipped)
    ops.adam_step(pb, g2, mb, vb, 1e-3 * c(2), 0.9, 0.999, 1e-8, nsq, 10.0)
    assert relerr(pa - w, pb - w) < 5e-4 and torch.equal(ma, mb)      # the moves are ~1e-3 of f32 values ~1
    # the scale itself
    ninf = torch.full((1,), float("inf"), dtype=torch.float32, device=DEV)
    ops.loss_scale_update(ls, ninf, interval=3)
    assert ls.cpu().tolist()[:4] == [512.0, 1.0 / 512.0, 0.0, 1.0]
    for k in range(3):
        ops.loss_scale_update(ls, nsq, interval=3)
    assert ls.cpu().tolist()[:4] == [1024.0, 1.0 / 1024.0, 0.0, 1.0]
    ops.loss_scale_update(ls, nsq, interval=3)
    assert ls.cpu().tolist()[2] == 1.0
    for k in range(40):                                              # clamps at both ends
        ops.loss_scale_update(ls, ninf, interval=3, min_scale=1.0)
    assert ls.cpu().tolist()[0] == 1.0
    for k in range(40 * 3):
        ops.loss_scale_update(ls, nsq, interval=3, max_scale=65536.0)
    assert ls.cpu().tolist()[0] == 65536.0


def test_f16_training_recovers_from_overflow():
    """An absurd initial scale overflows the first backward passes: those steps must leave the
    parameters untouched, the scale must come down by itself, and training must then proceed."""
    import types
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    V, E, C, H, W, B = 300, 128, 18, 32, 32, 2
    g = torch.Generator().manual_seed(5)
    grid = torch.zeros((B, H, W), dtype=torch.int32)
    occ = torch.rand((B, H, W), generator=g) < 0.15
    grid[occ] = torch.randint(1, V, (int(occ.sum()),), generator=g, dtype=torch.int32)
    labels = torch.zeros((B, H, W), dtype=torch.int32)
    fg = occ & (torch.rand((B, H, W), generator=g) < 0.3)
    labels[fg] = torch.randint(1, C, (int(fg.sum()),), generator=g, dtype=torch.int32)
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=B)
    net = CUTIERes(V, C, p)
    net.build_loss()
    eng = Engine(net, DEV, dtype="f16", seed=0, loss_scale=2.0 ** 40)
    tr = Trainer(eng, learning_rate=1e-3)
    before = eng.params_flat.clone()
    out = tr.step({"grid_table": grid, "gt_classes": labels})
    assert not np.isfinite(out["global_norm"])
    assert torch.equal(eng.params_flat, before)                      # skipped
    S, skipped = eng.loss_scale_value()
    assert S == 2.0 ** 39 and skipped == 1
    losses = []
    for k in range(60):
        out = tr.step({"grid_table": grid, "gt_classes": labels})
        losses.append(out["total_loss"])
    S, skipped = eng.loss_scale_value()
    print("scale settled at 2^%d after %d skipped steps; loss %.4f -> %.4f" % (int(np.log2(S)), skipped, losses[0], losses[-1]))
    assert 1 < skipped < 40 and np.isfinite(out["global_norm"])
    assert not torch.equal(eng.params_flat, before)
    assert np.isfinite(losses[-1]) and losses[-1] < losses[0]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_batch_scatter_add(ops, dt):
    """backward of batch_gather: most cells point at row 0 (token-less cells), some collide elsewhere,
    out-of-range indices clamp exactly as the forward does."""
    g = torch.Generator().manual_seed(14)
    B, n_cells, n_src, C = 3, 1500, 700, 32
    idx = torch.randint(0, n_src, (B, n_cells), generator=g, dtype=torch.int32)
    idx[torch.rand((B, n_cells), generator=g) < 0.7] = 0
    idx[0, 3], idx[0, 4], idx[1, 7] = -5, n_src + 9, idx[1, 8]
    dout = torch.randn((B, n_cells, C), generator=g).to(TDT[dt])
    wide = torch.zeros((B, n_cells, C + 32), dtype=TDT[dt], device=DEV)
    wide[..., :C] = dout.to(DEV)
    acc = torch.zeros((B, n_src, C), dtype=torch.float32, device=DEV)
    ops.batch_scatter_add(wide[..., :C], idx.to(DEV), acc)
    ref = torch.zeros((B, n_src, C), dtype=torch.float64)
    for b in range(B):
        ref[b].index_add_(0, idx[b].clamp(0, n_src - 1).long(), dout[b].double())
    assert relerr(acc, ref) < 1e-5


# ------------------------------------------------------------------------------------ GHM (use_ghm = 1)
def test_ghm_weights_match_calc_ghm_weights(ops):
    """cutie_ghm_weights against the restated calc_ghm_weights (main_train_json.py:86-119): bin edges hit
    exactly, gradients beyond the last edge, momentum state carried over two calls."""
    g = torch.Generator().manual_seed(31)
    n, C, bins = 4000, 18, 30
    logits = torch.randn((n, C), generator=g) * 0.8 + 0.3
    logits[:200] = torch.randint(0, bins + 1, (200, C), generator=g).float() / bins       # on the edges (as far as f32 can be)
    logits[200:260] = 1.0 + 1e-6                                                          # the last edge + eps
    logits[260:300] = 2.5                                                                 # outside every bin
    labels = torch.randint(0, C, (n,), generator=g, dtype=torch.int32)
    pad = torch.zeros((n, C + 2))
    pad[:, :C] = logits
    ld = pad.to(DEV)
    for momentum in (0.0, 0.75):
        state = ops.ghm_state(bins, DEV)
        acc = None
        for step in range(2):
            w = torch.zeros((n, C), dtype=torch.float32, device=DEV)
            z = torch.zeros((n, C), dtype=torch.float32, device=DEV)
            shift = 0.05 * step
            ops.ghm_weights(ld[:, :C] + shift if step else ld[:, :C], labels.to(DEV), C, bins, momentum, 1e-6, state, w, z)
            ref, acc = co.calc_ghm_weights((logits + shift if step else logits).reshape(1, 1, n, C), labels.reshape(1, 1, n),
                                           bins, momentum, 1e-6, acc)
            ref = ref.reshape(n, C).float()
            assert torch.equal(w.cpu(), ref), (momentum, step, float((w.cpu() - ref).abs().max()))
            assert torch.equal(z.cpu(), (logits + shift if step else logits) * ref)
    with pytest.raises(Exception):
        ops.ghm_state(300, DEV)
