"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's
token -> grid positional mapping.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (cutie_amd.*) never does; it calls the HIP kernel ``cutie_grid_build``.

Follows /root/reference/data_loader_json.py:437-577 (``DataLoader._positional_mapping``):
  :499-504  box_y/box_x, v_c/h_c in Python double arithmetic, row/col by int() truncation
  :507      stable sort by (row, [v_c, h_c], dict_id)
  :512-513  first free slot at or right of col (a cell holding id 0 is free)
  :518-529  strategy 0: drop the token when the row tail is full
  :530-557  strategy 1: shift the row left by one from its first free cell (labels, ps_x,
            ps_y are NOT shifted), then write at col = cols-1
  :559-563  crop the 2*cols working buffer to cols
Pinned against tests/golden/faktura1_grids.npz and synthetic_grids.npz, which were produced
by the reference's own code (tests/golden/make_golden.py).

Deviations, stated: the int8 label store wraps modulo 256 like numpy 1.15 (requirements.txt:20)
did; numpy 2 raises instead. Tokens whose row/col fall outside the grid raise ValueError
(the reference raises IndexError or writes into the cropped half of its buffer).
Strategy 2 (grid expansion) is restated here for completeness; the HIP kernel covers 0 and 1.
"""
import numpy as np

__all__ = ["token_keys", "positional_mapping_doc", "positional_mapping"]


def token_keys(box, extent, rows, cols):
    """(row, col, v_c, h_c, box_x, box_y) for one token - data_loader_json.py:499-504."""
    x_left, y_top, x_right, y_bottom = box
    left, top, right, bottom = extent
    box_y = y_top + (y_bottom - y_top) / 2
    box_x = x_left
    v_c = (y_top - top + (y_bottom - y_top) / 2) / (bottom - top)
    h_c = (x_left - left + (x_right - x_left) / 2) / (right - left)
    row = int(rows * v_c)
    col = int(cols * h_c)
    return row, col, v_c, h_c, box_x, box_y


def _wrap_int8(v):
    return int(np.array(v, dtype=np.int64).astype(np.int8))


def positional_mapping_doc(tokens, rows, cols, strategy, num_classes):
    """One document.

    tokens: list of dicts with keys box [xl,yt,xr,yb], extent [l,t,r,b], dict_id,
            class_id, entity_id  (order = reading order of the loader).
    Returns dict(grid int32 [rows, cols], labels int8, ps_x, ps_y int32,
                 order = token indices in placement order,
                 placed = per placement (row, col) or (-1, -1) when dropped,
                 cols_out)
    """
    cols_e = 2 * cols
    grid = np.zeros([rows, cols_e], dtype=np.int32)
    label = np.zeros([rows, cols_e], dtype=np.int8)
    ps_x = np.zeros([rows, cols_e], dtype=np.int32)
    ps_y = np.zeros([rows, cols_e], dtype=np.int32)

    recs = []
    for idx, t in enumerate(tokens):
        row, col, v_c, h_c, box_x, box_y = token_keys(t["box"], t["extent"], rows, cols)
        if row < 0 or row >= rows or col < 0 or col >= cols:
            raise ValueError("token %d maps outside the %dx%d grid: (%d,%d)" % (idx, rows, cols, row, col))
        recs.append((row, v_c, h_c, int(t["dict_id"]), idx, col, box_x, box_y))
    # python's list.sort is stable; ties beyond (row, v_c, h_c, dict_id) keep input order
    recs.sort(key=lambda r: (r[0], r[1], r[2], r[3]))

    order, placed = [], []
    for (row, v_c, h_c, dict_id, idx, col, box_x, box_y) in recs:
        t = tokens[idx]
        ecid = int(t["entity_id"]) * num_classes + int(t["class_id"])
        while col < cols and grid[row, col] != 0:
            col += 1
        order.append(idx)
        if strategy == 0:
            if col == cols:
                placed.append((-1, -1))
                continue
            grid[row, col] = dict_id
            label[row, col] = _wrap_int8(ecid)
            placed.append((row, col))
        elif strategy in (1, 2):
            if col == cols:
                col -= 1
                ptr = 0
                while ptr < cols and grid[row, ptr] != 0:
                    ptr += 1
                if ptr == cols:
                    grid[row, :-1] = grid[row, 1:].copy()
                else:
                    grid[row, ptr:-1] = grid[row, ptr + 1:].copy()
            if strategy == 2:
                while col < cols_e and grid[row, col] != 0:
                    col += 1
                if col > cols:
                    cols = col
            grid[row, col] = dict_id
            label[row, col] = _wrap_int8(ecid)
            ps_x[row, col] = int(box_x)
            ps_y[row, col] = int(box_y)      # float -> int32 store truncates toward zero
            placed.append((row, col))
        else:
            raise ValueError("unknown positional mapping strategy %r" % (strategy,))
    return {"grid": grid[:, :cols].copy(), "labels": label[:, :cols].copy(),
            "ps_x": ps_x[:, :cols].copy(), "ps_y": ps_y[:, :cols].copy(),
            "order": order, "placed": placed, "cols_out": cols}


def positional_mapping(docs, rows, cols, strategy, num_classes):
    """Batch of documents -> stacked arrays [B, rows, cols] (strategies 0/1: cols is fixed)."""
    outs = []
    for d in docs:
        o = positional_mapping_doc(d, rows, cols, strategy, num_classes)
        if strategy == 2:
            cols = o["cols_out"]      # the reference carries the grown cols into later docs
        outs.append(o)
    if strategy == 2:
        return outs
    return {"grid": np.stack([o["grid"] for o in outs]),
            "labels": np.stack([o["labels"] for o in outs]),
            "ps_x": np.stack([o["ps_x"] for o in outs]),
            "ps_y": np.stack([o["ps_y"] for o in outs]),
            "order": [o["order"] for o in outs], "placed": [o["placed"] for o in outs]}


def mapids_from_placement(tokens, order, placed, cols, num_classes):
    """label_mapid / bbox_mapid as the reference builds them (data_loader_json.py:459-460,
    :472,:489,:528,:556): bbox_id = 1-based reading-order index; label_mapid is appended in
    reading order BEFORE sorting; bbox_mapid is keyed row*cols+col in placement order."""
    label_mapid = [[] for _ in range(num_classes)]
    for i, t in enumerate(tokens):
        label_mapid[int(t["class_id"])].append(i + 1)
    bbox_mapid = {}
    for idx, (r, c) in zip(order, placed):
        if r >= 0:
            bbox_mapid[r * cols + c] = idx + 1
    return label_mapid, bbox_mapid
