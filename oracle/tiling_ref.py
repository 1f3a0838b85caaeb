"""ORACLE (test infrastructure) -- independent restatement of the reference's inference tile plan, crop windows, stitch
and recombination (TensorFlow/Prediction.py:259-310, :384-441, :443-481), written per axis with numpy index vectors.
Integer arithmetic: the product must match this BIT-EXACTLY.  It is itself pinned to tests/golden/tiling_golden.json, the
outputs of the reference's own lines executed by tests/golden/make_tiling_golden.py (tests/test_tiling_golden.py); further
known answers: SURVEY.md App. C.
"""
import math

import numpy as np


def _effective(height, width, tile, overlap):
    """Prediction.py:259-266: frames smaller than a tile shrink the tile and keep the overlap ratio."""
    short = min(height, width)
    if short < 16:
        raise Exception('The image needs to have at least a side length of 16 pixels.')
    if short < tile:
        tile, overlap = short, int(short * (overlap / tile))
    return tile, overlap


def _axis_origins(extent, tile, overlap):
    """Prediction.py:269-302 for one axis: count and lower bounds of the tiles."""
    step = tile - 2 * overlap
    count = math.ceil((extent - 2 * overlap - 2 * step) / step) + 2
    lo = np.arange(count) * step
    if count > 1:
        lo[-1] = extent - tile          # the last tile is flush with the far border
    return count, lo


def plan(height, width, tile_size=128, tile_overlap_size=14):
    """Returns (tile_size, overlap, height_count, width_count, windows) where windows[h][w] =
    (lower_height, upper_height, lower_width, upper_width)."""
    t, o = _effective(height, width, tile_size, tile_overlap_size)
    hc, ys = _axis_origins(height, t, o)
    wc, xs = _axis_origins(width, t, o)
    windows = [[(int(y), int(y) + t, int(x), int(x) + t) for x in xs] for y in ys]
    return t, o, hc, wc, windows


def crop(index, count, extent, tile_size, tile_overlap_size):
    """Valid window of tile `index` along one axis, in tile coordinates (Prediction.py:396-425): the first tile keeps its near
    border, the last one supplies exactly what the tiles before it have not covered, interior tiles drop one overlap per side."""
    first, last = index == 0, index == count - 1
    lower = 0 if first else tile_overlap_size
    upper = tile_size if last else tile_size - tile_overlap_size
    if last and not first:
        covered = tile_overlap_size + (count - 1) * (tile_size - 2 * tile_overlap_size)
        lower = tile_size - (extent - covered)
    return lower, upper


def stitch(tiles, height, width, tile_size=128, tile_overlap_size=14):
    """tiles[h][w]: [T,T,C] arrays in row-major plan order -> [height,width,C] (Prediction.py:384-441)."""
    t, o, hc, wc, _ = plan(height, width, tile_size, tile_overlap_size)
    stripes = []
    for hi in range(hc):
        lh, uh = crop(hi, hc, height, t, o)
        row = []
        for wi in range(wc):
            lw, uw = crop(wi, wc, width, t, o)
            row.append(tiles[hi][wi][lh:uh, lw:uw])
        stripes.append(np.concatenate(row, 1) if len(row) > 1 else row[0])
    return np.concatenate(stripes, 0) if len(stripes) > 1 else stripes[0]


def recombine(p):
    """Prediction.py:443-481; p maps pass name -> [H,W,3] array."""
    def comb(n):
        return np.multiply(p[n + ' Color'], np.add(p[n + ' Direct'], p[n + ' Indirect']))
    image = np.add(comb('Diffuse'), comb('Glossy'))
    image = np.add(image, comb('Subsurface'))
    image = np.add(image, comb('Transmission'))
    image = np.add(image, p['Volume Direct'])
    image = np.add(image, p['Volume Indirect'])
    image = np.add(image, p['Environment'])
    image = np.add(image, p['Emission'])
    return image
