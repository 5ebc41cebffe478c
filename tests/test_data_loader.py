"""Host-side mirror of the reference loader against golden vectors that
the reference's own data_loader_json.py / tokenization.py produced (tests/golden/make_golden.py).
The GPU part (row a2 through DataLoader._positional_mapping -> cutie_grid_build) is marked gpu."""
import hashlib
import json
import os
import random
import shutil
import types

import numpy as np
import pytest

from cutie_amd import dictfile, tokenization
from cutie_amd.data_loader_json import DataLoader, is_number

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLD, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tok_fix():
    return _load("tokenizer_cases.json")


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, tok_fix):
    """dict/vocab.txt + dict/20000T_* (the subsets the fixture carries) + two copies of the sample
    document, laid out like the reference's working directory."""
    root = tmp_path_factory.mktemp("loader")
    os.makedirs(root / "dict")
    os.makedirs(root / "invoice_data")
    with open(root / "dict" / "vocab.txt", "w", encoding="utf-8") as f:
        f.write("\n".join(tok_fix["vocab_subset"]) + "\n")
    w2i = dict(tok_fix["word_to_index_subset"])
    dictfile.save_npy_dict(str(root / "dict" / "20000T_dictionary.npy"), {w: 1 for w in w2i})
    dictfile.save_npy_dict(str(root / "dict" / "20000T_word_to_index.npy"), w2i)
    dictfile.save_npy_dict(str(root / "dict" / "20000T_index_to_word.npy"), {i: w for w, i in w2i.items()})
    for k in (0, 1):
        shutil.copy(os.path.join(GOLD, "invoice", "Faktura1.pdf_0.json"),
                    root / "invoice_data" / ("Faktura1.pdf_%d.json" % k))
    return root


def make_params(root, **over):
    p = types.SimpleNamespace(
        doc_path=str(root / "invoice_data"), test_path=str(root / "no_such_dir"), use_cutie2=False,
        text_case=True, tokenize=True, load_dict_from_path=str(root / "dict" / "20000T"),
        dict_path=str(root / "dict" / "20000T"), vocab_file=str(root / "dict" / "vocab.txt"),
        rows_segment=72, cols_segment=72, augment_strategy=1, positional_mapping_strategy=1,
        rows_target=64, cols_target=64, rows_ulimit=80, cols_ulimit=80, fill_bbox=False,
        data_augmentation_extra=True, data_augmentation_dropout=1,
        data_augmentation_extra_rows=16, data_augmentation_extra_cols=16, batch_size=1, save_dict=False)
    for k, v in over.items():
        setattr(p, k, v)
    return p


@pytest.fixture(scope="module")
def loader(workdir):
    return DataLoader(make_params(workdir), update_dict=False, load_dictionary=True, data_split=0.75,
                      verbose=False)


# ------------------------------------------------------------------------------ tokenizer
def test_tokenizer_matches_reference_outputs(tok_fix, workdir):
    toks = {False: tokenization.FullTokenizer(str(workdir / "dict" / "vocab.txt"), do_lower_case=False),
            True: tokenization.FullTokenizer(str(workdir / "dict" / "vocab.txt"), do_lower_case=True)}
    assert len(tok_fix["cases"]) > 200
    for c in tok_fix["cases"]:
        tk = toks[c["lower"]]
        assert tk.basic_tokenizer.tokenize(c["text"]) == c["basic"], c["text"]
        assert tk.tokenize(c["text"]) == c["pieces"], c["text"]


def test_wordpiece_edge_cases(workdir):
    tk = tokenization.FullTokenizer(str(workdir / "dict" / "vocab.txt"), do_lower_case=False)
    assert tk.tokenize("") == []
    assert tk.tokenize(" \t\n") == []
    assert tk.tokenize("x" * 201) == ["[UNK]"]                  # > 200 characters
    assert tk.basic_tokenizer.tokenize("a\x00b�c") == ["abc"]   # NUL / U+FFFD dropped
    assert tk.basic_tokenizer.tokenize("a b") == ["a", "b"]     # Zs is a blank
    ids = tk.convert_tokens_to_ids(["[UNK]"])
    assert tk.convert_ids_to_tokens(ids) == ["[UNK]"]


def test_is_number():
    assert is_number("7") and is_number("1e3") and is_number("½") and is_number("٣")
    assert not is_number("a") and not is_number("") and not is_number("1,5") and not is_number("½½")


def test_dress_text_matches_reference(tok_fix, loader, workdir):
    toks = {True: tokenization.FullTokenizer(str(workdir / "dict" / "vocab.txt"), do_lower_case=False),
            False: tokenization.FullTokenizer(str(workdir / "dict" / "vocab.txt"), do_lower_case=True)}
    saved = (loader.text_case, loader.tokenizer, dict(loader.dictionary))
    try:
        for c in tok_fix["dress_text"]:
            loader.text_case = c["text_case"]
            loader.tokenizer = toks[c["text_case"]]
            assert loader._dress_text(c["text"], False) == c["out"], (c["text"], c["text_case"])
    finally:
        loader.text_case, loader.tokenizer, loader.dictionary = saved


def test_dress_text_update_dict_adds_and_counts(loader):
    saved = dict(loader.dictionary)
    try:
        loader.dictionary.pop("zz9", None)
        out = loader._dress_text("[UNK]", False)
        n0 = loader.dictionary["[UNK]"]
        loader._dress_text("中", False)      # a CJK character outside the vocab subset
        assert loader.dictionary["[UNK]"] == n0 + 1
        assert out
    finally:
        loader.dictionary = saved


def test_dress_bbox(loader):
    assert loader._dress_bbox([10, 20, 110, 60]) == (10, 20, 110, 60)
    assert loader._dress_bbox([[5, 7], [40, 6], [41, 30], [4, 31]]) == (4, 6, 41, 31)     # quadrilateral
    assert loader._dress_bbox([-3.5, -1, 9.9, 8.2]) == (0, 0, 9, 8)                      # clamp + truncation


# ------------------------------------------------------------------------------ a1: _collect_data
def test_collect_data_matches_reference(loader):
    gold = _load("faktura1_items.json")
    assert loader.num_classes == gold["num_classes"] and loader.classes == gold["classes"]
    for split, docs in (("training", loader.training_docs), ("validation", loader.validation_docs)):
        assert len(docs) == len(gold["docs"][split])
        for doc, gdoc in zip(docs, gold["docs"][split]):
            assert len(doc) == len(gdoc) == 366
            for item, g in zip(doc, gdoc):
                assert item[0] == g["file_name"] and item[1] == g["text"] and item[2] == g["word_id"]
                assert [int(v) for v in item[3]] == g["box"]
                assert [int(v) for v in item[4]] == g["extent"]
                assert (item[5], item[6]) == (g["max_rows"], g["max_cols"]) == (42, 63)
                assert loader.word_to_index[item[1]] == g["dict_id"]
                assert loader._dress_class(item[0], item[2], loader.training_labels) == (g["entity_id"], g["class_id"])
    assert loader.training_labels == gold["labels"]


def test_collect_data_skips_special_and_empty(loader):
    content = [{"id": 1, "bbox": [], "text": "x"}, {"id": 2, "bbox": [0, 0, 10, 10], "text": "*"},
               {"id": 3, "bbox": [0, 0, 10, 10], "text": "="}, {"id": 4, "bbox": [20, 5, 60, 15], "text": "-"}]
    saved = dict(loader.dictionary)
    try:
        (doc,) = loader._collect_data("f.jpg", content, False)
    finally:
        loader.dictionary = saved
    assert len(doc) == 1 and doc[0][2] == 4
    assert doc[0][4] == [18, 3, 62, 17] and doc[0][5:] == [1, 1]


def test_natural_cols_grows_until_every_line_fits(loader):
    # five boxes on one text line, all centred in the same place: 5 columns are needed
    boxes = [(100, 10, 120, 20)] * 5
    assert loader._natural_cols(boxes, (0, 0, 200, 30), 1, 1) == 5
    # spread evenly they need no more than their count either
    boxes = [(40 * i, 10, 40 * i + 30, 20) for i in range(5)]
    assert loader._natural_cols(boxes, (-2, 8, 192, 22), 1, 1) <= 5


def test_segment_grid_splits_wide_documents(workdir):
    ld = DataLoader(make_params(workdir, segment_grid=True, cols_segment=40), update_dict=False,
                    load_dictionary=True, data_split=0.75, verbose=False)
    # natural 63 columns > 40: the document is offered as a left and a right part of 40+ columns
    docs = ld.training_docs + ld.validation_docs
    assert len(docs) == 4
    assert {d[0][6] for d in docs} == {40}
    assert all(d[0][5] == 42 for d in docs)
    assert len(docs[0]) + len(docs[1]) >= 366


# ------------------------------------------------------------------------------ a3: shapes
def test_cal_rows_cols_matches_reference(loader):
    gold = _load("shapes.json")
    assert [loader.rows, loader.cols] == gold["loader_rows_cols"]
    docs = loader.training_docs + loader.validation_docs
    assert [[d[0][5], d[0][6]] for d in docs] == gold["natural"]
    random.seed(7)
    for rec in gold["augment_strategy1"]:
        out = loader._cal_rows_cols(loader.training_docs, extra_augmentation=True)
        assert list(out) == rec["out"]
    assert loader._cal_rows_cols(loader.training_docs) == (42, 63, 42, 63)
    loader.augment_strategy = 9
    try:
        with pytest.raises(Exception, match="unknown augment strategy"):
            loader._cal_rows_cols(loader.training_docs, extra_augmentation=True)
    finally:
        loader.augment_strategy = 1


def test_pack_documents(loader):
    gold = _load("faktura1_items.json")["docs"]["training"][0]
    tok_box, tok_id, tok_ecid, tok_cls, extent, doc_off = loader.pack_documents(loader.training_docs, loader.training_labels)
    assert doc_off.tolist() == [0, 366] and extent.tolist() == [gold[0]["extent"]]
    assert tok_box.tolist() == [g["box"] for g in gold]
    assert tok_id.tolist() == [g["dict_id"] for g in gold]
    assert tok_ecid.tolist() == [g["entity_id"] * 18 + g["class_id"] for g in gold]
    assert tok_cls.tolist() == [g["class_id"] for g in gold]


def test_pack_documents_caches_per_document_and_concatenates(loader):
    doc = loader.training_docs[0]
    first = loader.pack_documents([doc, doc[:40], []], loader.training_labels)
    again = loader.pack_documents([doc, doc[:40], []], loader.training_labels)
    for a, b in zip(first, again):
        assert np.array_equal(a, b) and a.dtype == np.int32
    tok_box, tok_id, tok_ecid, tok_cls, extent, doc_off = first
    assert doc_off.tolist() == [0, 366, 366 + 40, 366 + 40] and tok_box.shape == (406, 4) and extent.shape == (3, 4)
    assert np.array_equal(tok_id[366:], tok_id[:40]) and extent[2].tolist() == [0, 0, 0, 0]
    assert loader._packed_doc(doc, loader.training_labels) is loader._packed_doc(doc, loader.training_labels)
    empty = loader.pack_documents([], loader.training_labels)
    assert empty[0].shape == (0, 4) and empty[5].tolist() == [0]


def test_prefetch_keeps_order_passes_errors_and_stops():
    from cutie_amd.data_loader_json import _Prefetcher
    import threading
    calls = []

    def fetch():
        calls.append(threading.current_thread().name)
        if len(calls) == 4:
            return None                       # fetch_test_data's end-of-pass marker
        return len(calls)
    it = _Prefetcher(fetch, depth=2)
    assert list(it) == [1, 2, 3] and set(calls) == {"cutie-prefetch"}

    def boom():
        raise IndexError("token outside the grid")
    it = _Prefetcher(boom, depth=1)
    with pytest.raises(IndexError):
        next(it)
    n = [0]

    def forever():
        n[0] += 1
        return n[0]
    it = _Prefetcher(forever, depth=2)
    assert next(it) == 1 and next(it) == 2
    it.close()
    assert not it._t.is_alive()


def test_unknown_mapping_strategy_is_refused(workdir):
    ld = DataLoader(make_params(workdir, positional_mapping_strategy=3), update_dict=False,
                    load_dictionary=True, verbose=False)
    with pytest.raises(ValueError):
        ld._positional_mapping(ld.training_docs, ld.training_labels, 64, 64)


@pytest.mark.gpu
def test_strategy2_on_gpu_matches_the_reference(workdir):
    """the reference's default strategy: tests/golden/strategy2_cases.json holds the sha256 of what the
    reference's own _positional_mapping returns for it"""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    cases = {c["name"]: c for c in _load("strategy2_cases.json")["cases"]}
    ld = DataLoader(make_params(workdir, positional_mapping_strategy=2), update_dict=False,
                    load_dictionary=True, verbose=False)
    for rows, cols in [(64, 64), (42, 63), (48, 20), (64, 8)]:
        c = cases["faktura1_r%dc%d" % (rows, cols)]
        out = ld._positional_mapping(ld.training_docs[:1], ld.training_labels, rows, cols)
        assert out[6] == c["updated_cols_s2"] == cols
        assert sha(np.stack(out[0])[..., 0].astype(np.int32)[..., None]) == c["s2_grid_sha256"]
        assert sha(np.stack(out[1]).astype(np.int8)) == c["s2_labels_sha256"]
        assert sha(np.stack(out[7]).astype(np.int32)) == c["s2_ps_x_sha256"]
        assert sha(np.stack(out[8]).astype(np.int32)) == c["s2_ps_y_sha256"]


def test_dictionary_round_trip(workdir, tmp_path):
    """ctor with save_dict writes files the reader takes back (data_loader_json.py:107-110)"""
    for f in os.listdir(workdir / "dict"):
        shutil.copy(workdir / "dict" / f, tmp_path / f)
    p = make_params(workdir, save_dict=True, load_dict_from_path=str(tmp_path / "20000T"))
    ld = DataLoader(p, update_dict=False, load_dictionary=True, verbose=False)
    d, w2i, i2w = dictfile.load_dictionary_triplet(str(tmp_path / "20000T"))
    assert w2i == ld.word_to_index and d == ld.dictionary and i2w == ld.index_to_word
    assert dictfile.load_npy_classes(str(tmp_path / "20000T_classes.npy")) == ld.classes


# ------------------------------------------------------------------------------ a2 through the loader (GPU)
def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["r64c64s1", "r64c64s0", "r42c63s1", "r80c80s1", "r64c32s1", "r48c20s0", "r48c20s1"])
def test_positional_mapping_on_gpu_matches_reference(loader, key):
    meta = _load("faktura1_grids.json")[key]
    if "error" in meta:
        pytest.skip("the reference itself fails on this shape")
    npz = np.load(os.path.join(GOLD, "faktura1_grids.npz"))
    loader.pm_strategy = meta["strategy"]
    try:
        out = loader._positional_mapping(loader.training_docs, loader.training_labels, meta["rows"], meta["cols"])
    finally:
        loader.pm_strategy = 1
    grid, lab, bboxes, label_mapids, bbox_mapids, file_names, cols, psx, psy = out
    assert np.array(grid).shape == (1, meta["rows"], meta["cols"], 1)
    assert np.array_equal(np.array(grid), npz[key + "_grid"].reshape(1, meta["rows"], meta["cols"], 1))
    assert np.array_equal(np.array(lab), npz[key + "_labels"])
    assert np.array_equal(np.array(psx), npz[key + "_ps_x"]) and np.array_equal(np.array(psy), npz[key + "_ps_y"])
    assert _sha(np.array(grid)[..., 0].astype(np.int32)) == meta["grid_sha256"]
    assert label_mapids == meta["label_mapids"]
    assert [{int(k): v for k, v in m.items()} for m in bbox_mapids] == \
           [{int(k): v for k, v in m.items()} for m in meta["bbox_mapids"]]
    assert file_names == meta["file_names"] and cols == meta["updated_cols"]


@pytest.mark.gpu
def test_next_batch_and_fetch_on_gpu(workdir):
    import torch
    ld = DataLoader(make_params(workdir, batch_size=1), update_dict=False, load_dictionary=True, verbose=False)
    random.seed(3)
    b = ld.next_batch()
    rows, cols = b["shape"]
    assert 42 <= rows <= 80 and 63 <= cols <= 80
    assert b["grid_table"].shape == (1, rows, cols, 1) and b["grid_table"].dtype == np.int32
    assert b["gt_classes"].shape == (1, rows, cols) and b["gt_classes"].dtype == np.int8
    assert b["data_image"].size == 0 and b["ps_1d_indices"].size == 0
    assert set(b) >= {"grid_table", "gt_classes", "data_image", "ps_1d_indices", "bboxes", "label_mapids",
                      "bbox_mapids", "file_name", "shape"}
    dev = b["device"]
    assert dev["grid_table"].is_cuda and torch.equal(dev["grid_table"].cpu(), torch.from_numpy(b["grid_table"]))
    assert (b["grid_table"] != 0).sum() == 366
    v = ld.fetch_validation_data()
    assert v["shape"] == [64, 64]
    npz = np.load(os.path.join(GOLD, "faktura1_grids.npz"))
    assert np.array_equal(v["grid_table"], npz["r64c64s1_grid"].reshape(1, 64, 64, 1))
    assert ld.fetch_test_data() is None


@pytest.mark.gpu
def test_prefetched_batches_feed_the_train_step(workdir):
    """loader thread ahead of the step: batches built on the background thread are
    the ones a direct call builds, and the trainer consumes their device tensors"""
    import types
    import torch
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    ld = DataLoader(make_params(workdir, batch_size=1), update_dict=False, load_dictionary=True, verbose=False)
    p = types.SimpleNamespace(embedding_size=128, weight_decay=5e-4, hard_negative_ratio=3, batch_size=1)
    net = CUTIERes(20000, ld.num_classes, p)
    net.build_loss()
    tr = Trainer(Engine(net, "cuda:0", dtype="f16", seed=0), learning_rate=1e-3)
    it = ld.prefetch(depth=2)
    losses = []
    for k, b in zip(range(6), it):
        assert (b["grid_table"] != 0).sum() == 366 and b["device"]["grid_table"].is_cuda
        assert torch.equal(b["device"]["grid_table"].cpu(), torch.from_numpy(b["grid_table"]))
        out = tr.step({"grid_table": b["device"]["grid_table"], "gt_classes": b["device"]["gt_classes"]})
        losses.append(out["total_loss"])
    it.close()
    assert len(losses) == 6 and all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]


# ------------------------------------------------------------------ a14: CUTIE2 image input + sampler indices
def _synthetic_image(path):
    import re
    m = re.search(r"_(\d+)x(\d+)\.jpg$", path)
    if not m:
        return None
    h, w = int(m.group(1)), int(m.group(2))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return np.stack([(yy * 3 + xx) % 256, (yy + 2 * xx) % 256, (yy * xx) % 256], axis=-1).astype(np.uint8)


def test_positional_sampling_indices_match_reference(workdir):
    """ps_1d, shapes and padding of DataLoader._positional_sampling against the reference's own function
    (tests/golden/positional_sampling_cases.json, made by make_golden_sampling.py with an injected image
    reader: the integer half is pinned, pixel values are not - cv2 is absent on both sides)."""
    ld = DataLoader(make_params(workdir, image_reader=_synthetic_image), update_dict=False, load_dictionary=True,
                    verbose=False)
    cases = _load("positional_sampling_cases.json")["cases"]
    assert len(cases) == 5
    for c in cases:
        psx = [np.array(a, dtype=np.int32) for a in c["ps_x"]]
        psy = [np.array(a, dtype=np.int32) for a in c["ps_y"]]
        images, ps_1d = ld._positional_sampling("imgs", c["file_names"], psx, psy, c["cols"])
        assert len(images) == len(ps_1d) == c["n_returned"], c["name"]
        assert [list(im.shape) for im in images] == c["image_shapes"], c["name"]
        for got, ref in zip(ps_1d, c["ps_1d"]):
            assert got.dtype == np.int32 and got.tolist() == ref, c["name"]
        for im in images:
            assert im.min() >= -0.5 and im.max() <= 0.5          # (x - 127.5) / 255 (:607)
    # the missing image ends the batch early, as the reference's `break` does (:608-612)
    assert [c["n_returned"] for c in cases if c["name"] == "missing_second"] == [1]


def test_positional_sampling_needs_a_reader(workdir):
    ld = DataLoader(make_params(workdir), update_dict=False, load_dictionary=True, verbose=False)
    with pytest.raises(NotImplementedError):
        ld._positional_sampling("imgs", ["a_10x10.jpg"], [np.zeros((4, 4))], [np.zeros((4, 4))], 4)


@pytest.mark.gpu
def test_cutie2_batches_train_end_to_end(workdir):
    """use_cutie2 loader -> data_image / ps_1d_indices -> CUTIE2 train step (config 4 / 5 plumbing)"""
    import types
    from cutie_amd.model_cutie2_aspp import CUTIE2
    from cutie_amd.runtime import Engine
    from cutie_amd.train import Trainer
    reader = lambda path: _synthetic_image("x_1000x800.jpg")
    ld = DataLoader(make_params(workdir, batch_size=1, use_cutie2=True, image_reader=reader), update_dict=False,
                    load_dictionary=True, verbose=False)
    random.seed(5)
    b = ld.next_batch()
    rows, cols = b["shape"]
    assert b["data_image"].shape[0] == 1 and b["data_image"].shape[2] == cols and b["data_image"].shape[3] == 3
    assert b["ps_1d_indices"].shape == (1, rows * cols)
    assert int(b["ps_1d_indices"].max()) <= cols * b["data_image"].shape[1] - 2
    p = types.SimpleNamespace(embedding_size=128, weight_decay=5e-4, hard_negative_ratio=3, batch_size=1)
    net = CUTIE2(20000, ld.num_classes, p)
    net.build_loss()
    tr = Trainer(Engine(net, "cuda:0", dtype="f16", seed=0), learning_rate=1e-3)
    losses = [tr.step({"grid_table": b["device"]["grid_table"], "gt_classes": b["device"]["gt_classes"],
                       "data_image": b["data_image"], "ps_1d_indices": b["ps_1d_indices"]})["total_loss"]
              for _ in range(5)]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]


@pytest.mark.gpu
def test_config0_invoice_to_accuracy_on_device_softmax(workdir):
    """BASELINE.json configs[0] end to end (main_evaluate_json.py:78-104): the reference's sample invoice ->
    fetch_validation_data -> Engine.forward (V = 20000, C = 18) -> cal_accuracy on the DEVICE's softmax equals the same
    call on the oracle's softmax (utils.py:14-92).  The model oracle is parity unpinned against TensorFlow."""
    import torch
    from oracle import cutie_oracle as co
    from cutie_amd.model_cutie_aspp import CUTIERes
    from cutie_amd.runtime import Engine
    from cutie_amd.utils import cal_accuracy
    ld = DataLoader(make_params(workdir, batch_size=1), update_dict=False, load_dictionary=True, verbose=False)
    random.seed(5)
    data = ld.fetch_validation_data()
    V, C, E = max(20000, ld.num_words), ld.num_classes, 128
    assert V == 20000 and C == 18 and data["grid_table"].shape[1] >= 64 and data["grid_table"].shape[2] >= 64
    p = types.SimpleNamespace(embedding_size=E, weight_decay=5e-4, hard_negative_ratio=3, batch_size=1)
    net = CUTIERes(V, C, p)
    net.build_loss()
    params = co.init_params(V, E, C, seed=0)
    # an untrained network marks no class above 0.5 anywhere: widen the logits layer's kernel so that the normalised
    # logits spread and both sides mark cells (two softmax tensors are compared, not a prediction with the truth)
    params["cls_logits/weights"] = params["cls_logits/weights"] * 40.0
    L = co.aspp_forward(params, torch.from_numpy(data["grid_table"]).long())
    ref_sm = L["softmax"].numpy()
    want = cal_accuracy(ld, data["grid_table"], data["gt_classes"], ref_sm, data["label_mapids"], data["bbox_mapids"])
    marked = int((ref_sm[..., 1:] > 0.5).sum())
    assert marked > 0
    for dtype in ("f32", "f16"):
        eng = Engine(net, "cuda:0", dtype=dtype)
        eng.load_params(params)
        P = eng.forward(data["device"]["grid_table"])             # the device tensor the loader's kernel wrote
        eng.check_inputs(P)
        sm = eng.get(P, "softmax").cpu().numpy()
        got = cal_accuracy(ld, data["grid_table"], data["gt_classes"], sm, data["label_mapids"], data["bbox_mapids"])
        flips = int(((sm[..., 1:] > 0.5) != (ref_sm[..., 1:] > 0.5)).sum())
        print("config 0 %s: prevalence %.4f strict %.4f soft %.4f; oracle marks %d (cell, class) pairs, %d differ"
              % ((dtype,) + tuple(got[:3]) + (marked, flips)))
        if flips == 0:
            assert got[1:3] == want[1:3] and abs(got[0] - want[0]) < 1e-12 and got[3] == want[3]
        else:                                                     # a probability within rounding of the 0.5 threshold
            assert dtype != "f32" and flips <= max(2, marked // 50) and abs(got[0] - want[0]) < 0.05


def test_load_data_process_pool_equals_serial(workdir):
    """f3: the load-time front half (JSON parse, WordPiece, natural-shape search) on a process pool gives the serial
    result - items, labels, dictionary counts AND the insertion order of new words (update_dict=True) - which is itself
    pinned by the reference's own items (tests/golden/faktura1_items.json)."""
    many = workdir / "many"
    os.makedirs(many, exist_ok=True)
    src = os.path.join(GOLD, "invoice", "Faktura1.pdf_0.json")
    with open(src, encoding="utf-8") as f:
        doc = json.load(f)
    for k in range(12):          # distinct file ids, a few distinct new words per file
        d = json.loads(json.dumps(doc))
        d["global_attributes"]["file_id"] = "copy_%02d.jpg" % k
        d["text_boxes"][3]["text"] = "zzq%dx" % (k % 5)
        d["text_boxes"][7]["text"] = "Qq%d" % (k // 3)
        with open(many / ("copy_%02d.json" % k), "w", encoding="utf-8") as f:
            json.dump(d, f)
    out = {}
    for workers in (1, 3):
        for update in (False, True):
            p = make_params(workdir, doc_path=str(many), load_workers=workers, save_dict=False)
            ld = DataLoader(p, update_dict=update, load_dictionary=True, verbose=False)
            assert ld._pool_size(12) == workers
            out[(workers, update)] = (ld.training_docs + ld.validation_docs, ld.training_labels, list(ld.dictionary.items()),
                                      ld.training_doc_files)
    for update in (False, True):
        a, b = out[(1, update)], out[(3, update)]
        assert a[3] == b[3]                       # same file order
        assert a[0] == b[0] and a[1] == b[1]
        assert a[2] == b[2], "dictionary words / counts / insertion order differ"
    assert len(out[(3, True)][2]) > len(out[(3, False)][2])       # update_dict added the new words
