#!/usr/bin/env python3
"""Host front half of the loader: JSON parse + WordPiece + natural-shape search per document
(data_loader_json.py:632-658, 704-785; tokenization.py:300-359), serial against the process pool.  CPU only.

    python tools/front_bench.py [n_files] [workers ...]

Documents: copies of the reference's sample invoice (tests/golden/invoice, 366 tokens after WordPiece) under distinct
file ids.  Prints documents per second of DataLoader.load_data at each worker count."""
import json
import os
import shutil
import sys
import tempfile
import time
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from cutie_amd import dictfile  # noqa: E402
from cutie_amd.data_loader_json import DataLoader  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    counts = [int(a) for a in sys.argv[2:]] or [1, os.cpu_count() or 1]
    gold = os.path.join(REPO, "tests", "golden")
    root = tempfile.mkdtemp(prefix="front_bench_")
    try:
        os.makedirs(os.path.join(root, "dict"))
        os.makedirs(os.path.join(root, "docs"))
        with open(os.path.join(gold, "tokenizer_cases.json"), encoding="utf-8") as f:
            tok = json.load(f)
        with open(os.path.join(root, "dict", "vocab.txt"), "w", encoding="utf-8") as f:
            f.write("\n".join(tok["vocab_subset"]) + "\n")
        w2i = dict(tok["word_to_index_subset"])
        dictfile.save_npy_dict(os.path.join(root, "dict", "20000T_dictionary.npy"), {w: 1 for w in w2i})
        dictfile.save_npy_dict(os.path.join(root, "dict", "20000T_word_to_index.npy"), w2i)
        dictfile.save_npy_dict(os.path.join(root, "dict", "20000T_index_to_word.npy"), {i: w for w, i in w2i.items()})
        with open(os.path.join(gold, "invoice", "Faktura1.pdf_0.json"), encoding="utf-8") as f:
            doc = json.load(f)
        for k in range(n):
            doc["global_attributes"]["file_id"] = "doc_%05d.jpg" % k
            with open(os.path.join(root, "docs", "doc_%05d.json" % k), "w", encoding="utf-8") as f:
                json.dump(doc, f)
        for workers in counts:
            p = types.SimpleNamespace(
                doc_path=os.path.join(root, "docs"), test_path=os.path.join(root, "none"), use_cutie2=False, text_case=True,
                tokenize=True, load_dict_from_path=os.path.join(root, "dict", "20000T"), dict_path=os.path.join(root, "dict", "20000T"),
                vocab_file=os.path.join(root, "dict", "vocab.txt"), batch_size=1, save_dict=False, load_workers=workers)
            t0 = time.perf_counter()
            ld = DataLoader(p, update_dict=False, load_dictionary=True, verbose=False)
            dt = time.perf_counter() - t0
            docs = len(ld.training_docs) + len(ld.validation_docs)
            print("workers %2d: %d documents in %.2f s = %.1f docs/s (whole DataLoader construction, pool start-up included)"
                  % (workers, docs, dt, docs / dt), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
