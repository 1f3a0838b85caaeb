"""Generates tests/golden/tiling_golden.json by EXECUTING the reference's own integer code.

The reference's tiling / index code is plain Python that happens to live inside modules importing TensorFlow, so the modules
cannot be imported here.  This script reads the cited line ranges out of /root/reference AT GENERATION TIME, dedents them,
exec()s them over a grid of synthetic sizes and records what they computed:

  * TensorFlow/Prediction.py:259-310  halo-tiling plan of a frame (effective tile/overlap, counts, the tile grid)
  * TensorFlow/Prediction.py:384-441  crop + stitch of the per-tile predictions back into the frame
  * TensorFlow/Training.py:879-913    source_index_tuples (Python `random`, seeded)
  * TensorFlow/TFRecordsCreator.py:125-133  training-side tile grid (remainders dropped)

Nothing of the reference's text is embedded here or in the JSON: the committed fixture holds inputs and the outputs the reference
code produced.  Run in the build container only (/root/reference does not exist on the GPU box):
    python tests/golden/make_tiling_golden.py
"""
import json
import math
import os
import random
import sys
import textwrap
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/TensorFlow"
sys.path.insert(0, REF)
from Naming import Naming  # noqa: E402  (pure Python in the reference)

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_lines(fname, first, last, must_contain):
    """Source lines [first, last] (1-based, inclusive) of a reference file as a compiled code object.  `must_contain` are short
    identifiers expected on the first and last line -- a guard against the line numbers drifting, not a copy of the text."""
    with open(os.path.join(REF, fname)) as f:
        lines = f.read().split("\n")
    block = lines[first - 1:last]
    assert must_contain[0] in block[0] and must_contain[1] in block[-1], (fname, first, last, block[0], block[-1])
    return compile(textwrap.dedent("\n".join(block)), "%s:%d-%d" % (fname, first, last), "exec")


PLAN = ref_lines("Prediction.py", 259, 310, ("smaller_side_length", "tiled_features_grid"))
STITCH = ref_lines("Prediction.py", 384, 441, ("predictions", "prediction_name"))
TUPLES = ref_lines("Training.py", 879, 913, ("source_index_tuples", "return"))
TRAIN_TILES = ref_lines("TFRecordsCreator.py", 125, 133, ("tiles_x_count", "y2"))


def run_plan(height, width, tile_size, tile_overlap_size):
    yy, xx = np.meshgrid(np.arange(height, dtype=np.int32), np.arange(width, dtype=np.int32), indexing="ij")
    index_image = np.stack([yy, xx], -1)
    ns = {"height": height, "width": width, "tile_size": tile_size, "tile_overlap_size": tile_overlap_size,
          "features": {"f": index_image}, "math": math}
    try:
        exec(PLAN, ns)
    except Exception as e:          # the reference raises for frames smaller than 16 pixels
        return {"error": str(e)}
    T, o, hc, wc = ns["tile_size"], ns["tile_overlap_size"], ns["height_count"], ns["width_count"]
    grid = ns["tiled_features_grid"]
    case = {"tile": int(T), "overlap": int(o), "height_count": int(hc), "width_count": int(wc),
            "row_origins": [int(grid[i][0]["f"][0, 0, 0]) for i in range(hc)],
            "col_origins": [int(grid[0][j]["f"][0, 0, 1]) for j in range(wc)]}
    tiles_ok = all(grid[i][j]["f"].shape == (T, T, 2) for i in range(hc) for j in range(wc))
    case["all_tiles_full_size"] = bool(tiles_ok)
    if not tiles_ok:                # degenerate plans (the reference would fail later); recorded, not stitched
        return case

    # identity "network": every tile predicts its own (global y, global x, tile id)
    name = Naming.feature_prediction_name("X")
    pgrid = [[{name: np.concatenate([grid[i][j]["f"], np.full((T, T, 1), i * wc + j, np.int32)], -1)} for j in range(wc)] for i in range(hc)]
    fp = types.SimpleNamespace(name="X", load_data=True)
    arch = types.SimpleNamespace(feature_prediction_tuples=[types.SimpleNamespace(feature_predictions=[fp])])
    ns2 = {"architecture": arch, "Naming": Naming, "np": np, "tiled_features_grid": pgrid, "height_count": hc, "width_count": wc,
           "tile_size": T, "tile_overlap_size": o, "height": height, "width": width}
    try:
        exec(STITCH, ns2)
    except Exception as e:
        case["stitch_error"] = type(e).__name__
        return case
    out = ns2["predictions"][name]
    case["stitched_shape"] = [int(v) for v in out.shape[:2]]
    covers = out.shape[:2] == (height, width) and bool(np.array_equal(out[..., :2], index_image))
    case["stitch_is_identity"] = covers
    if covers:                      # which tile supplied each output row / column, and from where inside the tile
        rows, cols = out[:, 0, 2] // wc, out[0, :, 2] % wc
        rc, cc = [], []
        for i in range(hc):
            ys = np.nonzero(rows == i)[0]
            rc.append([int(ys[0] - case["row_origins"][i]), int(ys[-1] + 1 - case["row_origins"][i]), int(ys[0])] if len(ys) else None)
        for j in range(wc):
            xs = np.nonzero(cols == j)[0]
            cc.append([int(xs[0] - case["col_origins"][j]), int(xs[-1] + 1 - case["col_origins"][j]), int(xs[0])] if len(xs) else None)
        case["row_crops"], case["col_crops"] = rc, cc      # [lo, hi) in tile coordinates, offset in the stitched frame
        # every tile of a row band must contribute exactly that band (the crop is separable)
        sep = all(np.all(out[:, x, 2] // wc == rows) for x in (0, width - 1, width // 2)) and \
            all(np.all(out[y, :, 2] % wc == cols) for y in (0, height - 1, height // 2))
        case["separable"] = bool(sep)
    return case


def main():
    out = {"_generator": "tests/golden/make_tiling_golden.py",
           "_source": ["TensorFlow/Prediction.py:259-310", "TensorFlow/Prediction.py:384-441", "TensorFlow/Training.py:879-913",
                       "TensorFlow/TFRecordsCreator.py:125-133"]}
    sizes = [15, 16, 17, 23, 31, 63, 64, 65, 97, 99, 100, 101, 113, 127, 128, 129, 131, 199, 200, 201, 227, 228, 229, 256, 257, 300, 541, 1080]
    plans = []
    seen = set()

    def add(h, w, t, o):
        if (h, w, t, o) in seen:
            return
        seen.add((h, w, t, o))
        plans.append({"height": h, "width": w, "tile_size": t, "tile_overlap_size": o, "result": run_plan(h, w, t, o)})
    for t, o in ((128, 14), (64, 7), (128, 0), (32, 5), (100, 10), (128, 31), (48, 3)):
        for n in sizes:
            add(n, n, t, o)
            add(n, 300, t, o)
            add(131, n, t, o)
        for n in (t - 1, t, t + 1, 2 * t - 2 * o - 1, 2 * t - 2 * o, 2 * t - 2 * o + 1, 3 * t - 4 * o, 3 * t - 4 * o + 1):
            if n >= 8:
                add(n, n + 37, t, o)
    for h, w in ((1080, 1920), (540, 960), (720, 1280), (2160, 3840), (1087, 1931), (256, 256), (100, 300), (128, 128), (1080, 127)):
        add(h, w, 128, 14)
    out["plans"] = plans

    tuples = []
    for seed in (0, 1, 7):
        for s_ex in (1, 2, 3, 4, 8):
            for n_t in (1, 3, 8, 9, 17):
                for per_t in (1, 2, 3):
                    ns = {"random": random}
                    exec(TUPLES, ns)
                    random.seed(seed)
                    try:
                        it, req = ns["source_index_tuples"](s_ex, n_t, per_t)
                        res = {"index_tuples": it, "required_indices": req}
                    except Exception as e:
                        res = {"error": str(e)}
                    tuples.append({"seed": seed, "number_of_sources_per_example": s_ex, "number_of_source_index_tuples": n_t,
                                   "number_of_sources_per_target": per_t, "result": res})
    out["source_index_tuples"] = tuples

    ttiles = []
    for h, w, t in ((1080, 1920, 128), (1080, 1920, 64), (540, 960, 64), (128, 128, 128), (127, 500, 128), (200, 129, 64), (64, 63, 64), (333, 777, 100)):
        # the cited lines end inside the loop body (before the reference builds its feature dictionaries): record (x1,x2,y1,y2) per
        # iteration through a recording namespace; `self` carries the tile size
        rec = []

        class Rec(dict):
            def __setitem__(self, k, v):
                dict.__setitem__(self, k, v)
                if k == "y2":
                    rec.append([self["x1"], self["x2"], self["y1"], self["y2"]])
        ns = Rec({"height": h, "width": w, "self": types.SimpleNamespace(tiles_height_width=t)})
        exec(TRAIN_TILES, ns)
        ttiles.append({"height": h, "width": w, "tiles_height_width": t, "tiles_x_count": ns["tiles_x_count"], "tiles_y_count": ns["tiles_y_count"],
                       "tiles": rec})
    out["training_tiles"] = ttiles

    path = os.path.join(HERE, "tiling_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s: %d plans, %d tuple cases, %d training grids, %d bytes" % (path, len(plans), len(tuples), len(ttiles), os.path.getsize(path)))


if __name__ == "__main__":
    main()
