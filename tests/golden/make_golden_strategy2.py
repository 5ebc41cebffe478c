"""Generates tests/golden/strategy2_cases.json by running the REFERENCE's own
DataLoader._positional_mapping (data_loader_json.py:437-577) with positional_mapping_strategy 2
next to strategy 1 on the documents the other fixtures already hold.  Runs only in the build
container (needs /root/reference, imported exactly as make_golden.py does); the output is data:
per case the sha256 of the strategy-2 grid / labels / ps arrays, the returned column count, and
whether each equals the strategy-1 result.

Why: strategy 2 ("expand grid table if overlap", :541-550) looks for a free slot to the right of
column cols-1 AFTER the strategy-1 left shift (:532-539) has already freed that column (the shift
pulls in column `cols` of the 2*cols-wide buffer, which nothing ever writes), so the search stops
at once and `cols` never grows.  This script checks that claim against the reference itself.
"""
import json
import os
import random
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg._install_stubs()
    work = tempfile.mkdtemp(prefix="cutie_golden_s2_")
    os.makedirs(os.path.join(work, "dict"))
    os.makedirs(os.path.join(work, "invoice_data"))
    for f in os.listdir(os.path.join(mg.REF, "dict")):
        if f.startswith("20000T_") or f == "vocab.txt":
            shutil.copy(os.path.join(mg.REF, "dict", f), os.path.join(work, "dict", f))
    for f in os.listdir(os.path.join(mg.REF, "invoice_data")):
        if f.endswith(".json"):
            shutil.copy(os.path.join(mg.REF, "invoice_data", f), os.path.join(work, "invoice_data", f))
    os.chdir(work)
    sys.path.insert(0, mg.REF)
    sys.dont_write_bytecode = True
    import data_loader_json as ref
    np.load = mg._safe_np_load
    random.seed(0)
    loader = ref.DataLoader(mg.make_params(), update_dict=False, load_dictionary=True, data_split=0.75)

    out = {"cases": []}

    def compare(name, docs, labels, rows, cols):
        r1 = mg.run_mapping(loader, docs, labels, rows, cols, 1)
        r2 = mg.run_mapping(loader, docs, labels, rows, cols, 2)
        rec = {"name": name, "rows": rows, "cols": cols, "updated_cols_s1": r1["updated_cols"],
               "updated_cols_s2": r2["updated_cols"]}
        same = True
        for k in ("grid", "labels", "ps_x", "ps_y"):
            rec["s2_%s_sha256" % k] = mg.sha(r2[k])
            eq = r1[k].shape == r2[k].shape and bool(np.array_equal(r1[k], r2[k]))
            rec["%s_equal_to_s1" % k] = eq
            same = same and eq
        rec["mapids_equal_to_s1"] = (r1["label_mapids"] == r2["label_mapids"] and r1["bbox_mapids"] == r2["bbox_mapids"])
        rec["identical_to_s1"] = bool(same and rec["mapids_equal_to_s1"] and r1["updated_cols"] == r2["updated_cols"])
        out["cases"].append(rec)

    doc = loader.training_docs[:1]
    for (rows, cols) in [(64, 64), (42, 63), (80, 80), (64, 32), (48, 20), (64, 8), (16, 64)]:
        try:
            compare("faktura1_r%dc%d" % (rows, cols), doc, loader.training_labels, rows, cols)
        except Exception as e:          # the reference's own failure mode is data too
            out["cases"].append({"name": "faktura1_r%dc%d" % (rows, cols), "error": type(e).__name__})

    # the synthetic documents of synthetic_grids.json, rebuilt from the stored records
    with open(os.path.join(HERE, "synthetic_grids.json")) as f:
        syn = json.load(f)
    loader.word_to_index = mg._Identity()
    for case in syn["cases"]:
        table = {}
        docs = []
        for recs in case["docs"]:
            items = []
            for r in recs:
                table[(r["file_name"], r["word_id"])] = (r["entity_id"], r["class_id"])
                items.append([r["file_name"], r["dict_id"], r["word_id"], r["box"], r["extent"], 0, 0])
            docs.append(items)
        loader._dress_class = lambda file_name, word_id, labels, t=table: t[(file_name, word_id)]
        try:
            compare("synthetic_" + case["name"], docs, {}, case["rows"], case["cols"])
        except Exception as e:
            out["cases"].append({"name": "synthetic_" + case["name"], "error": type(e).__name__})

    with open(os.path.join(HERE, "strategy2_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    for c in out["cases"]:
        print(c["name"], c.get("identical_to_s1"), c.get("updated_cols_s2"), c.get("error"))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
