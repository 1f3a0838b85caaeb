"""ORACLE (test infrastructure) -- literal restatement of the reference's inference tile plan,
crop windows, stitch and recombination (TensorFlow/Prediction.py:259-311, :384-441, :443-481).
Integer arithmetic: the product must match this BIT-EXACTLY.  Known answers: SURVEY.md App. C.
"""
import math

import numpy as np


def plan(height, width, tile_size=128, tile_overlap_size=14):
    """Returns (tile_size, overlap, height_count, width_count, windows) where windows[h][w] =
    (lower_height, upper_height, lower_width, upper_width) -- Prediction.py:259-311."""
    smaller_side_length = min(height, width)
    if smaller_side_length < 16:
        raise Exception('The image needs to have at least a side length of 16 pixels.')
    if smaller_side_length < tile_size:
        ratio = tile_overlap_size / tile_size
        tile_size = smaller_side_length
        tile_overlap_size = int(tile_size * ratio)
    iteration_delta = tile_size - (2 * tile_overlap_size)
    width_count = width - (2 * tile_overlap_size) - (2 * iteration_delta)
    width_count = width_count / iteration_delta
    width_count = math.ceil(width_count) + 2
    height_count = height - (2 * tile_overlap_size) - (2 * iteration_delta)
    height_count = height_count / iteration_delta
    height_count = math.ceil(height_count) + 2
    windows = [[None for _ in range(width_count)] for _ in range(height_count)]
    for height_index in range(height_count):
        if height_index == 0:
            lower_height, upper_height = 0, tile_size
        elif height_index == height_count - 1:
            upper_height = height
            lower_height = upper_height - tile_size
        else:
            lower_height = height_index * iteration_delta
            upper_height = lower_height + tile_size
        for width_index in range(width_count):
            if width_index == 0:
                lower_width, upper_width = 0, tile_size
            elif width_index == width_count - 1:
                upper_width = width
                lower_width = upper_width - tile_size
            else:
                lower_width = width_index * iteration_delta
                upper_width = lower_width + tile_size
            windows[height_index][width_index] = (lower_height, upper_height, lower_width, upper_width)
    return tile_size, tile_overlap_size, height_count, width_count, windows


def crop(index, count, extent, tile_size, tile_overlap_size):
    """Valid window of tile `index` along one axis, in tile coordinates -- Prediction.py:396-425."""
    lower, upper = 0, tile_size
    if index != 0 and index != count - 1:
        lower = tile_overlap_size
        upper = upper - tile_overlap_size
    elif index == 0 and index == count - 1:
        pass
    elif index == 0:
        upper = upper - tile_overlap_size
    else:
        existing = tile_overlap_size + ((count - 1) * (tile_size - (2 * tile_overlap_size)))
        remaining = extent - existing
        lower = upper - remaining
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
