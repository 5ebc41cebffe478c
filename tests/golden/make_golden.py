#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE loader.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py

What it does
  * imports /root/reference/data_loader_json.py with two stub modules registered in
    sys.modules (``tensorflow`` exposing gfile.GFile = open(utf-8); empty ``cv2``) - the
    loader half of the reference uses neither after import;
  * re-points ``numpy.load`` at cutie_amd.dictfile's non-executing reader before the
    DataLoader is constructed, so the reference's dict/*.npy files are never unpickled;
  * works in a scratch directory holding copies of dict/20000T_* + dict/vocab.txt and
    invoice_data/*.json, because the ctor writes the dict files back next to themselves;
  * writes DATA only: inputs (token items, bbox ints, ids, labels) and the reference's
    outputs (grids, label grids, ps_x/ps_y, mapids, tokenizer splits, natural shapes).

Outputs
  faktura1_items.json       a1: _collect_data output for both sample docs + dict ids/classes
  faktura1_grids.npz        a2: _positional_mapping at several rows x cols x strategy
  synthetic_grids.json/npz  a2: seeded synthetic docs incl. crowded rows (shift-left), id 0
  tokenizer_cases.json      tokenization.FullTokenizer splits + the vocab subset they need
  shapes.json               a3: _cal_rows_cols / natural shapes
  accuracy_cases.json/npz   f2: utils.cal_accuracy on the Faktura1 64x64 grid and seeded model outputs
  invoice/Faktura1.pdf_0.json  copy of the reference's own sample input (data file)
"""
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from cutie_amd import dictfile  # noqa: E402


def _install_stubs():
    tf = types.ModuleType("tensorflow")
    gfile = types.SimpleNamespace()
    gfile.GFile = lambda path, mode="r": open(path, mode, encoding="utf-8")
    tf.gfile = gfile
    sys.modules["tensorflow"] = tf
    sys.modules["cv2"] = types.ModuleType("cv2")


def _safe_np_load(path, *a, **k):
    """numpy.load replacement used while the reference code runs: object arrays go through
    the non-executing reader and come back as 0-d object arrays (so ``.item()`` works)."""
    try:
        return _np_load_plain(path, allow_pickle=False)
    except ValueError:
        out = np.empty((), dtype=object)
        out[()] = dictfile.load_npy_dict(path)
        return out


_np_load_plain = np.load


def make_params(**over):
    p = types.SimpleNamespace(
        doc_path="invoice_data", test_path="no_such_dir", use_cutie2=False,
        text_case=True, tokenize=True, update_dict=False, load_dict=True,
        load_dict_from_path="dict/20000T", dict_path="dict/20000T",
        rows_segment=72, cols_segment=72, augment_strategy=1,
        positional_mapping_strategy=1, rows_target=64, cols_target=64,
        rows_ulimit=80, cols_ulimit=80, fill_bbox=False,
        data_augmentation_extra=True, data_augmentation_dropout=1,
        data_augmentation_extra_rows=16, data_augmentation_extra_cols=16,
        batch_size=1)
    for k, v in over.items():
        setattr(p, k, v)
    return p


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def jsonable_items(loader, doc, labels):
    """One record per token: everything _positional_mapping reads (data_loader_json.py:462-470)."""
    out = []
    for it in doc:
        file_name, text, word_id, box, extent, mr, mc = it
        ent, cls = loader._dress_class(file_name, word_id, labels)
        out.append({"file_name": file_name, "text": text, "word_id": word_id,
                    "box": [int(v) for v in box], "extent": [int(v) for v in extent],
                    "max_rows": int(mr), "max_cols": int(mc),
                    "dict_id": int(loader.word_to_index[text]),
                    "entity_id": int(ent), "class_id": int(cls)})
    return out


def run_mapping(loader, docs, labels, rows, cols, strategy):
    loader.pm_strategy = strategy
    (grid, lab, bboxes, label_mapids, bbox_mapids, file_names, ucols, psx, psy) = \
        loader._positional_mapping(docs, labels, rows, cols)
    return {
        "grid": np.stack([np.asarray(g) for g in grid]).astype(np.int32),
        "labels": np.stack([np.asarray(g) for g in lab]).astype(np.int8),
        "ps_x": np.stack(psx).astype(np.int32), "ps_y": np.stack(psy).astype(np.int32),
        "label_mapids": label_mapids,
        "bbox_mapids": [{int(k): int(v) for k, v in m.items()} for m in bbox_mapids],
        "file_names": file_names, "updated_cols": int(ucols),
    }


def synthetic_doc(rng, name, n_tok, page_w, page_h, n_lines, zero_frac, num_classes, vocab_hi,
                  crowd=False):
    """Token items in the _collect_data output format (data_loader_json.py:828-829)."""
    items, labels_by_class = [], {}
    left, top, right, bottom = 9999, 9999, 0, 0
    raw = []
    for t in range(n_tok):
        line = rng.randrange(n_lines)
        h = page_h // (n_lines + 2)
        y_top = 20 + line * h + rng.randrange(0, 3)
        y_bot = y_top + max(4, h - rng.randrange(2, 6))
        w = rng.randrange(10, 120) if not crowd else rng.randrange(4, 24)
        x_left = rng.randrange(10, page_w - w - 10)
        if crowd and rng.random() < 0.7:
            x_left = page_w - w - 10 - rng.randrange(0, 40)   # pile up at the right edge
        x_right = x_left + w
        r = rng.random()
        if r < zero_frac:
            did = 0
        elif r < zero_frac + 0.3:
            did = 1
        else:
            did = rng.randrange(2, vocab_hi)
        raw.append((x_left, y_top, x_right, y_bot, did))
        # same extent update rule as data_loader_json.py:719-722
        if x_left < left:
            left = x_left - 2
        if y_top < top:
            top = y_top - 2
        if x_right > right:
            right = x_right + 2
        if y_bot > bottom:
            bottom = y_bot + 2
    for t, (x_left, y_top, x_right, y_bot, did) in enumerate(raw):
        wid = t + 1
        if rng.random() < 0.2:
            cls = rng.randrange(1, num_classes)
            labels_by_class.setdefault(cls, []).append(wid)
        items.append([name, did, wid, [x_left, y_top, x_right, y_bot],
                      [left, top, right, bottom], 0, 0])
    return items, labels_by_class


def main():
    _install_stubs()
    work = tempfile.mkdtemp(prefix="cutie_golden_")
    os.makedirs(os.path.join(work, "dict"))
    os.makedirs(os.path.join(work, "invoice_data"))
    for f in os.listdir(os.path.join(REF, "dict")):
        if f.startswith("20000T_") or f == "vocab.txt":
            shutil.copy(os.path.join(REF, "dict", f), os.path.join(work, "dict", f))
    for f in os.listdir(os.path.join(REF, "invoice_data")):
        if f.endswith(".json"):
            shutil.copy(os.path.join(REF, "invoice_data", f), os.path.join(work, "invoice_data", f))
    os.chdir(work)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import data_loader_json as ref  # the reference module (sets np.load = allow_pickle wrapper)
    np.load = _safe_np_load          # ... which we replace before anything is loaded
    import tokenization as ref_tok

    random.seed(0)
    params = make_params()
    loader = ref.DataLoader(params, update_dict=False, load_dictionary=True, data_split=0.75)
    os.makedirs(os.path.join(HERE, "invoice"), exist_ok=True)
    shutil.copy(os.path.join(REF, "invoice_data", "Faktura1.pdf_0.json"),
                os.path.join(HERE, "invoice", "Faktura1.pdf_0.json"))

    # ---------------------------------------------------------------- a1: items
    docs = {"training": loader.training_docs, "validation": loader.validation_docs}
    items_out = {"num_words": int(loader.num_words), "num_classes": int(loader.num_classes),
                 "classes": list(loader.classes), "docs": {}}
    for split, dd in docs.items():
        items_out["docs"][split] = [jsonable_items(loader, d, loader.training_labels) for d in dd]
    items_out["labels"] = loader.training_labels
    with open(os.path.join(HERE, "faktura1_items.json"), "w", encoding="utf-8") as f:
        json.dump(items_out, f, ensure_ascii=False, indent=0)

    # ---------------------------------------------------------------- a2: Faktura1 grids
    npz, meta = {}, {}
    doc = loader.training_docs[:1]
    for (rows, cols) in [(64, 64), (42, 63), (80, 80), (64, 32), (48, 20), (64, 8), (16, 64)]:
        for strategy in (0, 1):
            key = "r%dc%ds%d" % (rows, cols, strategy)
            try:
                r = run_mapping(loader, doc, loader.training_labels, rows, cols, strategy)
            except Exception as e:  # record the reference's own failure mode as data
                meta[key] = {"error": type(e).__name__}
                continue
            for k in ("grid", "labels", "ps_x", "ps_y"):
                npz[key + "_" + k] = r[k]
            meta[key] = {"rows": rows, "cols": cols, "strategy": strategy,
                         "grid_sha256": sha(r["grid"]), "labels_sha256": sha(r["labels"]),
                         "label_mapids": r["label_mapids"], "bbox_mapids": r["bbox_mapids"],
                         "updated_cols": r["updated_cols"], "file_names": r["file_names"]}
    np.savez_compressed(os.path.join(HERE, "faktura1_grids.npz"), **npz)
    with open(os.path.join(HERE, "faktura1_grids.json"), "w") as f:
        json.dump(meta, f)

    # ---------------------------------------------------------------- a2: synthetic docs
    rng = random.Random(1234)
    syn_meta, syn_npz = {"cases": []}, {}
    # the reference looks ids up through word_to_index[text]; give it an identity mapping on ints
    saved_w2i = loader.word_to_index
    loader.word_to_index = _Identity()
    num_classes = loader.num_classes
    case_specs = [
        # name, n_docs, n_tok, page_w, page_h, lines, zero_frac, rows, cols, strategy, crowd
        ("plain",   3, 300, 1500, 2100, 40, 0.00, 64, 64, 1, False),
        ("zeros",   2, 300, 1500, 2100, 40, 0.15, 64, 64, 1, False),
        ("crowd1",  3, 500, 1500, 2100, 12, 0.05, 32, 16, 1, True),
        ("crowd0",  3, 500, 1500, 2100, 12, 0.05, 32, 16, 0, True),
        ("tiny",    2,   5,  300,  200,  2, 0.00,  4,  4, 1, False),
        ("full",    1, 900, 1500, 2100,  6, 0.00,  8,  8, 1, True),
        ("wide",    2, 600, 3000, 1000, 30, 0.02, 40, 128, 1, False),
        ("single",  1,   1,  300,  200,  1, 0.00, 64, 64, 1, False),
    ]
    for (name, n_docs, n_tok, pw, ph, lines, zf, rows, cols, strategy, crowd) in case_specs:
        sdocs, slabels = [], {}
        for d in range(n_docs):
            fname = "%s_%d.jpg" % (name, d)
            items, lbc = synthetic_doc(rng, fname, n_tok, pw, ph, lines, zf, num_classes, 11000, crowd)
            sdocs.append(items)
            slabels[fname] = {loader.classes[c]: [{"key_id": [], "value_id": ids,
                                                   "key_text": "", "value_text": ""}]
                              for c, ids in lbc.items()}
            # a second entity of one class -> entity_class_id >= num_classes
            if name == "plain" and d == 0 and lbc:
                c0 = sorted(lbc)[0]
                slabels[fname][loader.classes[c0]].append(
                    {"key_id": [], "value_id": [n_tok], "key_text": "", "value_text": ""})
        r = run_mapping(loader, sdocs, slabels, rows, cols, strategy)
        case = {"name": name, "rows": rows, "cols": cols, "strategy": strategy, "docs": []}
        for items in sdocs:
            recs = []
            for it in items:
                ent, cls = loader._dress_class(it[0], it[2], slabels)
                recs.append({"file_name": it[0], "word_id": it[2], "box": it[3], "extent": it[4],
                             "dict_id": int(it[1]), "entity_id": int(ent), "class_id": int(cls)})
            case["docs"].append(recs)
        for k in ("grid", "labels", "ps_x", "ps_y"):
            syn_npz[name + "_" + k] = r[k]
        case["label_mapids"] = r["label_mapids"]
        case["bbox_mapids"] = r["bbox_mapids"]
        case["updated_cols"] = r["updated_cols"]
        syn_meta["cases"].append(case)
    loader.word_to_index = saved_w2i
    syn_meta["num_classes"] = num_classes
    np.savez_compressed(os.path.join(HERE, "synthetic_grids.npz"), **syn_npz)
    with open(os.path.join(HERE, "synthetic_grids.json"), "w") as f:
        json.dump(syn_meta, f)

    # ---------------------------------------------------------------- tokenizer cases
    with open(os.path.join(REF, "invoice_data", "Faktura1.pdf_0.json"), encoding="utf-8") as f:
        raw = json.load(f)
    texts = sorted({tb["text"] for tb in raw["text_boxes"]})
    texts += ["unaffable", "HELLO wOrld", "naïve café", "12.345,67", "北京 2020", "a\tb c",
              "x" * 201, "", "   ", "élève-École", "#hash@tag!", "Zażółć gęślą jaźń"]
    tok_cased = ref_tok.FullTokenizer("dict/vocab.txt", do_lower_case=False)
    tok_uncased = ref_tok.FullTokenizer("dict/vocab.txt", do_lower_case=True)
    cases, needed = [], set(["[UNK]", "[PAD]"])
    for t in texts:
        for lower, tk in ((False, tok_cased), (True, tok_uncased)):
            basic = tk.basic_tokenizer.tokenize(t)
            pieces = tk.tokenize(t)
            cases.append({"text": t, "lower": lower, "basic": basic, "pieces": pieces})
            # vocab subset that keeps greedy longest-match identical: every vocab entry that is a
            # prefix-candidate (a substring of a basic token, with/without the ## marker)
            for b in basic:
                for s in range(len(b)):
                    for e in range(s + 1, len(b) + 1):
                        sub = b[s:e] if s == 0 else "##" + b[s:e]
                        if sub in tk.vocab:
                            needed.add(sub)
    dressed = []
    for t in texts:
        for tc in (True, False):
            loader.text_case = tc
            loader.tokenizer = tok_cased if tc else tok_uncased
            # the loader maps numeric characters to '0' before splitting (:892-895): the vocab
            # subset must also cover the substrings of those strings
            mapped = "".join("0" if ref.is_number(c) else c for c in (t if tc else t.lower()))
            for b in loader.tokenizer.basic_tokenizer.tokenize(mapped):
                for s in range(len(b)):
                    for e in range(s + 1, len(b) + 1):
                        sub = b[s:e] if s == 0 else "##" + b[s:e]
                        if sub in tok_cased.vocab:
                            needed.add(sub)
            before = dict(loader.dictionary)
            dressed.append({"text": t, "text_case": tc, "out": loader._dress_text(t, False)})
            loader.dictionary = before
    loader.text_case, loader.tokenizer = True, tok_cased
    # keep file order of the real vocab so ids stay monotone
    vocab_subset = [w for w in tok_cased.vocab if w in needed]
    w2i_subset = {}
    for d in dressed:
        for w in d["out"]:
            if w in saved_w2i:
                w2i_subset[w] = int(saved_w2i[w])
    with open(os.path.join(HERE, "tokenizer_cases.json"), "w", encoding="utf-8") as f:
        json.dump({"cases": cases, "vocab_subset": vocab_subset, "dress_text": dressed,
                   "dictionary_words": sorted(w for w in saved_w2i
                                              if any(w in d["out"] for d in dressed)),
                   "word_to_index_subset": w2i_subset}, f, ensure_ascii=False)

    # ---------------------------------------------------------------- a3: shapes
    shapes = {"natural": [[int(d[0][5]), int(d[0][6])] for d in loader.training_docs + loader.validation_docs],
              "loader_rows_cols": [int(loader.rows), int(loader.cols)]}
    random.seed(7)
    loader.data_augmentation_extra = True
    aug = []
    for _ in range(8):
        state = random.getstate()
        g0 = random.gauss(0, loader.da_extra_rows * loader.encoding_factor)
        g1 = random.gauss(0, loader.da_extra_cols * loader.encoding_factor)
        random.setstate(state)
        aug.append({"gauss": [g0, g1],
                    "out": [int(v) for v in loader._cal_rows_cols(loader.training_docs, extra_augmentation=True)]})
    shapes["augment_strategy1"] = aug
    with open(os.path.join(HERE, "shapes.json"), "w") as f:
        json.dump(shapes, f)

    # ---------------------------------------------------------------- f2: utils.cal_accuracy
    import utils as ref_utils
    loader.pm_strategy = 1
    r = run_mapping(loader, loader.training_docs[:1], loader.training_labels, 64, 64, 1)
    grid_table = r["grid"].reshape(1, 64, 64, 1)
    gt = r["labels"].astype(np.int64).reshape(1, 64, 64)
    rs = np.random.RandomState(5)
    acc_cases = []
    for case, (flip, noise) in enumerate([(0.0, 0.02), (0.15, 0.03), (0.5, 0.04), (0.03, 0.0)]):
        # a softmax-like output: one-hot of the labels, a share of the labelled cells flipped to
        # another class, uniform noise, renormalised
        C = loader.num_classes
        lab = gt.copy()
        mask = (rs.rand(*lab.shape) < flip) & (grid_table[..., 0] != 0)
        lab[mask] = rs.randint(0, C, size=int(mask.sum()))
        out = np.eye(C)[lab] + noise * rs.rand(1, 64, 64, C)
        out = (out / out.sum(-1, keepdims=True)).astype(np.float32)
        prev, strict, soft, res = ref_utils.cal_accuracy(loader, grid_table, gt, out, r["label_mapids"], r["bbox_mapids"])
        acc_cases.append({"flip": flip, "noise": noise, "prevalence": float(prev), "accuracy_strict": float(strict),
                          "accuracy_soft": float(soft), "res": res.decode("utf-8")})
        npz_acc = {"model_output_%d" % case: out}
        syn_acc = npz_acc if case == 0 else {**syn_acc, **npz_acc}
    np.savez_compressed(os.path.join(HERE, "accuracy_cases.npz"), grid_table=grid_table, gt_classes=gt, **syn_acc)
    grid_ids = sorted(set(int(v) for v in grid_table.reshape(-1)))
    with open(os.path.join(HERE, "accuracy_cases.json"), "w", encoding="utf-8") as f:
        json.dump({"cases": acc_cases, "label_mapids": r["label_mapids"], "bbox_mapids": r["bbox_mapids"],
                   "classes": list(loader.classes),
                   "index_to_word": {str(i): loader.index_to_word[i] for i in grid_ids}}, f, ensure_ascii=False)

    print("golden fixtures written to", HERE)
    g = npz["r64c64s1_grid"]
    print("faktura1 64x64 grid sha256", sha(g), "labels", sha(npz["r64c64s1_labels"]))
    shutil.rmtree(work, ignore_errors=True)


class _Identity(dict):
    def __getitem__(self, k):
        return k

    def __contains__(self, k):
        return True


if __name__ == "__main__":
    main()
