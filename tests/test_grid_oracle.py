"""Pins oracle/grid_oracle.py to the reference's own outputs (tests/golden, produced by
tests/golden/make_golden.py from /root/reference/data_loader_json.py:437-577)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import grid_oracle as go

# hashes of the reference's Faktura1 64x64 grid / label grid
GRID_SHA = "f0a48dabd7fb03a53eff5c5968854c65ec7b518f3143bf722fb3c0cb65844479"
LABEL_SHA = "ebe7f40cec2ced0fca7f3e8201172e7be5feed4491978519b3517e8a735ca27f"


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def faktura(golden_dir):
    with open(os.path.join(golden_dir, "faktura1_items.json"), encoding="utf-8") as f:
        items = json.load(f)
    with open(os.path.join(golden_dir, "faktura1_grids.json")) as f:
        meta = json.load(f)
    npz = np.load(os.path.join(golden_dir, "faktura1_grids.npz"))
    return items, meta, npz


def test_faktura_items_shape(faktura):
    items, _, _ = faktura
    doc = items["docs"]["training"][0]
    assert len(doc) == 366
    assert sum(1 for t in doc if t["dict_id"] == 1) == 132      # [UNK]
    assert doc[0]["max_rows"] == 42 and doc[0]["max_cols"] == 63
    assert items["num_classes"] == 18


def test_faktura_64x64_hash(faktura):
    items, _, npz = faktura
    doc = items["docs"]["training"][0]
    out = go.positional_mapping([doc], 64, 64, 1, items["num_classes"])
    assert _sha(out["grid"][..., None].astype(np.int32)) == GRID_SHA
    assert _sha(out["labels"].astype(np.int8)) == LABEL_SHA
    g = out["grid"][0]
    assert [(0, c, int(g[0, c])) for c in (0, 2, 4, 5, 8)] == \
        [(0, 0, 1), (0, 2, 1), (0, 4, 1), (0, 5, 27), (0, 8, 1)]
    assert int((out["labels"] != 0).sum()) == 81


def test_faktura_all_shapes_and_strategies(faktura):
    items, meta, npz = faktura
    doc = items["docs"]["training"][0]
    nc = items["num_classes"]
    checked = 0
    for key, m in meta.items():
        if "error" in m:
            continue
        out = go.positional_mapping([doc], m["rows"], m["cols"], m["strategy"], nc)
        np.testing.assert_array_equal(out["grid"], npz[key + "_grid"][..., 0], err_msg=key)
        np.testing.assert_array_equal(out["labels"], npz[key + "_labels"], err_msg=key)
        if m["strategy"] == 1:          # strategy 0 never writes ps_x/ps_y (:526-529)
            np.testing.assert_array_equal(out["ps_x"], npz[key + "_ps_x"], err_msg=key)
            np.testing.assert_array_equal(out["ps_y"], npz[key + "_ps_y"], err_msg=key)
        lm, bm = go.mapids_from_placement(doc, out["order"][0], out["placed"][0], m["cols"], nc)
        assert lm == m["label_mapids"][0], key
        assert bm == {int(k): v for k, v in m["bbox_mapids"][0].items()}, key
        checked += 1
    assert checked >= 12


def test_synthetic_cases(golden_dir):
    with open(os.path.join(golden_dir, "synthetic_grids.json")) as f:
        meta = json.load(f)
    npz = np.load(os.path.join(golden_dir, "synthetic_grids.npz"))
    nc = meta["num_classes"]
    for case in meta["cases"]:
        name = case["name"]
        out = go.positional_mapping(case["docs"], case["rows"], case["cols"], case["strategy"], nc)
        np.testing.assert_array_equal(out["grid"], npz[name + "_grid"][..., 0], err_msg=name)
        np.testing.assert_array_equal(out["labels"], npz[name + "_labels"], err_msg=name)
        if case["strategy"] == 1:
            np.testing.assert_array_equal(out["ps_x"], npz[name + "_ps_x"], err_msg=name)
            np.testing.assert_array_equal(out["ps_y"], npz[name + "_ps_y"], err_msg=name)
        for d, doc in enumerate(case["docs"]):
            lm, bm = go.mapids_from_placement(doc, out["order"][d], out["placed"][d], case["cols"], nc)
            assert lm == case["label_mapids"][d], name
            assert bm == {int(k): v for k, v in case["bbox_mapids"][d].items()}, name


def test_strategy2_restatement_matches_the_reference(faktura, golden_dir):
    """tests/golden/strategy2_cases.json: the reference's strategy-2 outputs (sha256) and its own verdict
    that they equal strategy 1's on every fixture document; the oracle restates :541-550 literally."""
    items, _, _ = faktura
    with open(os.path.join(golden_dir, "strategy2_cases.json")) as f:
        cases = {c["name"]: c for c in json.load(f)["cases"]}
    assert len(cases) == 15 and all(c["identical_to_s1"] for c in cases.values())
    doc = items["docs"]["training"][0]
    for name, c in cases.items():
        if not name.startswith("faktura1_"):
            continue
        o = go.positional_mapping([doc], c["rows"], c["cols"], 2, items["num_classes"])[0]
        assert o["cols_out"] == c["updated_cols_s2"] == c["cols"]
        assert _sha(o["grid"][None, ..., None].astype(np.int32)) == c["s2_grid_sha256"], name
        assert _sha(o["labels"][None].astype(np.int8)) == c["s2_labels_sha256"], name
        assert _sha(o["ps_x"][None].astype(np.int32)) == c["s2_ps_x_sha256"], name
        assert _sha(o["ps_y"][None].astype(np.int32)) == c["s2_ps_y_sha256"], name
    with open(os.path.join(golden_dir, "synthetic_grids.json")) as f:
        meta = json.load(f)
    for case in meta["cases"]:
        c = cases["synthetic_" + case["name"]]
        outs = go.positional_mapping(case["docs"], case["rows"], case["cols"], 2, meta["num_classes"])
        assert all(o["cols_out"] == case["cols"] for o in outs)
        assert _sha(np.stack([o["grid"] for o in outs])[..., None].astype(np.int32)) == c["s2_grid_sha256"], case["name"]
        assert _sha(np.stack([o["labels"] for o in outs]).astype(np.int8)) == c["s2_labels_sha256"], case["name"]
        assert _sha(np.stack([o["ps_x"] for o in outs]).astype(np.int32)) == c["s2_ps_x_sha256"], case["name"]


def test_label_overflow_present_in_fixture(golden_dir):
    """a second entity of one class gives labels >= num_classes."""
    npz = np.load(os.path.join(golden_dir, "synthetic_grids.npz"))
    assert int(npz["plain_labels"].max()) >= 18


def test_out_of_grid_raises():
    tok = {"box": [0, 0, 10, 10], "extent": [0, 0, 5, 5], "dict_id": 3, "class_id": 0, "entity_id": 0}
    with pytest.raises(ValueError):
        go.positional_mapping([[tok]], 4, 4, 1, 18)
