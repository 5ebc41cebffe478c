"""Synthetic documents of the shapes SURVEY.md section 8(d) names (no datasets offline):
per document T ~ U{200..600} OCR tokens on a 1500 x 2100 page in ~40 text lines, box width
proportional to token length; ids: 35 % [UNK] (1), the rest Zipf(1.1) over 2..max_id;
labels: 80 % class 0, otherwise U{1..C-1}.  Returned in the packed layout cutie_grid_build reads."""
import numpy as np


def make_documents(n_docs, num_classes=18, max_id=11126, seed=1234, page_w=1500, page_h=2100, n_lines=40,
                   t_lo=200, t_hi=600):
    boxes, ids, ecids, extents, offs = [], [], [], [], [0]
    line_h = page_h // (n_lines + 2)
    for d in range(n_docs):
        rng = np.random.RandomState(seed + d)
        T = int(rng.randint(t_lo, t_hi + 1))
        line = rng.randint(0, n_lines, size=T)
        tok_len = rng.randint(1, 13, size=T)
        w = 9 * tok_len + rng.randint(0, 6, size=T)
        x_left = (rng.rand(T) * (page_w - w - 20)).astype(np.int64) + 10
        y_top = 20 + line * line_h + rng.randint(0, 3, size=T)
        y_bot = y_top + line_h - rng.randint(2, 6, size=T)
        x_right = x_left + w
        z = rng.zipf(1.1, size=T)
        tid = 2 + (z - 1) % (max_id - 1)
        tid = np.where(rng.rand(T) < 0.35, 1, tid)
        cls = np.where(rng.rand(T) < 0.8, 0, rng.randint(1, num_classes, size=T))
        boxes.append(np.stack([x_left, y_top, x_right, y_bot], axis=1))
        ids.append(tid)
        ecids.append(cls)       # entity 0
        # extents as data_loader_json.py:719-722 leaves them (first-minimum minus 2 / maximum plus 2)
        left = top = 9999
        right = bottom = 0
        for xl, yt, xr, yb in boxes[-1]:
            if xl < left: left = xl - 2
            if yt < top: top = yt - 2
            if xr > right: right = xr + 2
            if yb > bottom: bottom = yb + 2
        extents.append([left, top, right, bottom])
        offs.append(offs[-1] + T)
    i32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.int32)
    return {"tok_box": i32(np.concatenate(boxes)), "tok_id": i32(np.concatenate(ids)),
            "tok_ecid": i32(np.concatenate(ecids)), "doc_extent": i32(extents), "doc_off": i32(offs)}


def as_token_dicts(packed):
    """The same documents as lists of token dicts (what oracle.grid_oracle consumes)."""
    docs = []
    off = packed["doc_off"]
    for d in range(len(off) - 1):
        doc = []
        for t in range(off[d], off[d + 1]):
            doc.append({"box": [int(v) for v in packed["tok_box"][t]], "extent": [int(v) for v in packed["doc_extent"][d]],
                        "dict_id": int(packed["tok_id"][t]), "class_id": int(packed["tok_ecid"][t]), "entity_id": 0})
        docs.append(doc)
    return docs
