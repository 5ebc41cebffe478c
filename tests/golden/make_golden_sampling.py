"""Generates tests/golden/positional_sampling_cases.json from the REFERENCE's own
DataLoader._positional_sampling (data_loader_json.py:579-630), run in the build container.

cv2 is not installed, so the two calls the function makes into it are given minimal stand-ins on the
stub module make_golden.py already registers: `imread` returns a synthetic uint8 image whose size
is encoded in the file name, `resize` is a nearest-neighbour index map.  What the fixture pins is
therefore the INTEGER half of the function - the float32 scaling, flooring, row-major index and the
`max_w*h - 2` clamp of `ps_1d` (:587-603) - plus the output shapes and the zero padding to the
tallest image (:617-628).  Pixel values are not pinned (they would be the stand-in's, not cv2's).
"""
import json
import os
import random
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def synthetic_image(path):
    m = re.search(r"_(\d+)x(\d+)\.jpg$", path)
    if not m:
        return None
    h, w = int(m.group(1)), int(m.group(2))
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    return np.stack([(yy * 3 + xx) % 256, (yy + 2 * xx) % 256, (yy * xx) % 256], axis=-1).astype(np.uint8)


def nn_resize(image, size):
    w, h = size
    yy = np.minimum((np.arange(h) * image.shape[0] / h).astype(np.int64), image.shape[0] - 1)
    xx = np.minimum((np.arange(w) * image.shape[1] / w).astype(np.int64), image.shape[1] - 1)
    return image[yy][:, xx]


def main():
    mg._install_stubs()
    sys.modules["cv2"].imread = synthetic_image
    sys.modules["cv2"].resize = nn_resize
    work = tempfile.mkdtemp(prefix="cutie_golden_ps_")
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

    rng = np.random.RandomState(7)
    out = {"cases": []}
    specs = [  # name, cols (= image width after resize), rows, [(image h, image w), ...]
        ("one_down", 64, 64, [(2100, 1500)]),
        ("two_mixed", 64, 64, [(1000, 800), (1754, 1240)]),
        ("upscale", 128, 40, [(60, 100)]),
        ("ragged3", 80, 72, [(1200, 1000), (900, 1000), (1500, 1001)]),
        ("missing_second", 64, 64, [(1000, 800), None, (1000, 800)]),
    ]
    for name, cols, rows, sizes in specs:
        file_names, psx, psy = [], [], []
        for k, hw in enumerate(sizes):
            if hw is None:
                file_names.append("%s_%d_missing.jpg" % (name, k))
                hw = (1000, 800)
            else:
                file_names.append("%s_%d_%dx%d.jpg" % (name, k, hw[0], hw[1]))
            occ = rng.rand(rows, cols) < 0.15
            x = np.where(occ, rng.randint(0, hw[1], (rows, cols)), 0).astype(np.int32)
            y = np.where(occ, rng.randint(0, hw[0], (rows, cols)), 0).astype(np.int32)
            x[0, 0], y[0, 0] = hw[1] - 1, hw[0] - 1          # the last pixel: exercises the clamp
            psx.append(x)
            psy.append(y)
        images, ps_1d = loader._positional_sampling("imgs", file_names, psx, psy, cols)
        out["cases"].append({
            "name": name, "cols": cols, "rows": rows, "file_names": file_names,
            "ps_x": [a.tolist() for a in psx], "ps_y": [a.tolist() for a in psy],
            "n_returned": len(images),
            "image_shapes": [list(im.shape) for im in images],
            "pad_rows_all_zero": [bool(np.all(im[nn_h:] == 0)) for im, nn_h in
                                  zip(images, [int(hw[0] * (cols / hw[1])) for hw in sizes if hw is not None][:len(images)])],
            "ps_1d": [np.asarray(p).astype(np.int64).tolist() for p in ps_1d],
            "ps_1d_dtype": [str(np.asarray(p).dtype) for p in ps_1d],
        })
    with open(os.path.join(HERE, "positional_sampling_cases.json"), "w") as f:
        json.dump(out, f)
    for c in out["cases"]:
        print(c["name"], c["n_returned"], c["image_shapes"], c["ps_1d_dtype"], [max(p) for p in c["ps_1d"]])
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
