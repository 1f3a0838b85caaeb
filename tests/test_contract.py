"""Naming / render-pass contract against fixtures generated from the REFERENCE's own pure-Python modules
(tests/golden/make_naming_golden.py), and the integer tiling contract against the literal restatement."""
import json
import os

import numpy as np
import pytest

from deepdenoiser_amd.naming import Naming
from deepdenoiser_amd.render_passes import RenderPasses, RenderPassesUsage
from deepdenoiser_amd import tiling
from oracle import tiling_ref

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "naming_golden.json")))


def test_pass_constants():
    for k, v in GOLD["pass_constants"].items():
        assert getattr(RenderPasses, k) == v, k


def test_render_pass_functions():
    for fn, table in GOLD["render_pass_functions"].items():
        for name, expected in table.items():
            if expected == "__AttributeError__":
                with pytest.raises(AttributeError):
                    getattr(RenderPasses, fn)(name)
            else:
                assert getattr(RenderPasses, fn)(name) == expected, (fn, name)


def test_render_passes_usage_order():
    for case in GOLD["usage_cases"]:
        assert RenderPassesUsage(**{f: True for f in case["flags"]}).render_passes() == case["passes"]
    with pytest.raises(TypeError):
        RenderPassesUsage(use_nonexistent=True)


def test_naming_calls():
    assert len(GOLD["naming_calls"]) > 400
    for fn, args, kwargs, expected in GOLD["naming_calls"]:
        assert getattr(Naming, fn)(*args, **kwargs) == expected, (fn, args, kwargs)


# ---- tiling: SURVEY Appendix C known answers + bit-exact agreement with the literal restatement
def test_tile_plan_known_answers():
    p = tiling.tile_plan(1080, 1920)
    assert (p.rows.count, p.cols.count, p.count) == (11, 19, 209)
    assert p.rows.origins == (0, 100, 200, 300, 400, 500, 600, 700, 800, 900, 952)
    assert p.cols.origins[-3:] == (1600, 1700, 1792)
    assert p.rows.crops[-1] == (62, 128) and p.cols.crops[-1] == (22, 128)
    p = tiling.tile_plan(540, 960)
    assert (p.rows.count, p.cols.count) == (6, 10) and p.rows.origins[-2:] == (400, 412) and p.cols.origins[-2:] == (800, 832)
    assert p.rows.crops[-1] == (102, 128) and p.cols.crops[-1] == (82, 128)
    p = tiling.tile_plan(256, 256)
    assert p.rows.origins == (0, 100, 128) and p.rows.crops[-1] == (86, 128)
    p = tiling.tile_plan(128, 128)
    assert p.count == 1 and p.rows.crops == ((0, 128),)
    p = tiling.tile_plan(100, 300)
    assert (p.tile, p.overlap, p.rows.count, p.cols.count) == (100, 10, 1, 4)
    assert p.cols.origins == (0, 80, 160, 200) and p.cols.crops[-1] == (50, 100)
    with pytest.raises(Exception):
        tiling.tile_plan(15, 200)


@pytest.mark.parametrize("h,w,t,o", [(1080, 1920, 128, 14), (540, 960, 128, 14), (256, 256, 128, 14), (128, 128, 128, 14),
                                      (100, 300, 128, 14), (333, 517, 128, 14), (129, 130, 128, 14), (64, 64, 128, 14),
                                      (720, 1280, 96, 10), (17, 4000, 128, 14), (2160, 3840, 256, 20), (200, 200, 64, 0)])
def test_tile_plan_matches_reference_restatement(h, w, t, o):
    p = tiling.tile_plan(h, w, t, o)
    rt, ro, hc, wc, windows = tiling_ref.plan(h, w, t, o)
    assert (p.tile, p.overlap, p.rows.count, p.cols.count) == (rt, ro, hc, wc)
    for hi in range(hc):
        for wi in range(wc):
            lh, uh, lw, uw = windows[hi][wi]
            assert (p.rows.origins[hi], p.cols.origins[wi]) == (lh, lw) and uh - lh == rt and uw - lw == rt
        assert p.rows.crops[hi] == tiling_ref.crop(hi, hc, h, rt, ro)
    for wi in range(wc):
        assert p.cols.crops[wi] == tiling_ref.crop(wi, wc, w, rt, ro)
    # every pixel exactly once
    cover = np.zeros((h, w), dtype=np.int32)
    for hi in range(hc):
        for wi in range(wc):
            (a, b), (c, d) = p.rows.crops[hi], p.cols.crops[wi]
            y0, x0 = p.rows.origins[hi] + a, p.cols.origins[wi] + c
            assert (y0, x0) == (p.rows.offsets[hi], p.cols.offsets[wi])
            cover[y0:y0 + b - a, x0:x0 + d - c] += 1
    assert (cover == 1).all()


def test_stitch_roundtrip_restatement():
    rng = np.random.default_rng(0)
    h, w = 300, 420
    img = rng.standard_normal((h, w, 3)).astype(np.float32)
    t, o, hc, wc, windows = tiling_ref.plan(h, w)
    tiles = [[img[a:b, c:d] for (a, b, c, d) in row] for row in windows]
    assert np.array_equal(tiling_ref.stitch(tiles, h, w), img)
