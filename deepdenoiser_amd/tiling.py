"""Inference halo-tiling plan, crop windows and stitch offsets (integer contract).

Bit-exact with the reference's Prediction.py:259-311 (plan) and :396-427 (crops); tiles are
produced and consumed row-major (:325-326, :380-382).  Known answers: SURVEY.md Appendix C,
pinned in tests/test_tiling.py against oracle/tiling_ref.py (the literal restatement).
"""

import math
from dataclasses import dataclass
from typing import List, Tuple


@dataclass(frozen=True)
class AxisPlan:
    extent: int
    tile: int
    overlap: int
    count: int
    origins: Tuple[int, ...]          # tile origin in image coordinates
    crops: Tuple[Tuple[int, int], ...]  # valid window [lo, hi) in tile coordinates
    offsets: Tuple[int, ...]          # where crop `i` lands in the stitched image

    @property
    def delta(self):
        return self.tile - 2 * self.overlap


@dataclass(frozen=True)
class TilePlan:
    height: int
    width: int
    tile: int
    overlap: int
    rows: AxisPlan
    cols: AxisPlan

    @property
    def count(self):
        return self.rows.count * self.cols.count

    def windows(self) -> List[Tuple[int, int]]:
        """Row-major list of (y0, x0) tile origins."""
        return [(y, x) for y in self.rows.origins for x in self.cols.origins]


def effective_tile(height, width, tile_size=128, tile_overlap_size=14):
    """Prediction.py:259-266: shrink the tile (keeping the overlap ratio) for small frames."""
    smaller = min(height, width)
    if smaller < 16:
        raise Exception("The image needs to have at least a side length of 16 pixels.")
    if smaller < tile_size:
        ratio = tile_overlap_size / tile_size
        tile_size = smaller
        tile_overlap_size = int(tile_size * ratio)
    return tile_size, tile_overlap_size


def _axis(extent, tile, overlap):
    delta = tile - 2 * overlap
    # float division then ceil, exactly as the reference does (:272-278)
    count = math.ceil((extent - 2 * overlap - 2 * delta) / delta) + 2
    origins, crops = [], []
    for i in range(count):
        if i == 0:
            origins.append(0)
        elif i == count - 1:
            origins.append(extent - tile)
        else:
            origins.append(i * delta)
        first, last = i == 0, i == count - 1
        if first and last:
            crops.append((0, tile))
        elif first:
            crops.append((0, tile - overlap))
        elif last:
            remaining = extent - (overlap + (count - 1) * delta)
            crops.append((tile - remaining, tile))
        else:
            crops.append((overlap, tile - overlap))
    offsets, pos = [], 0
    for lo, hi in crops:
        offsets.append(pos)
        pos += hi - lo
    return AxisPlan(extent, tile, overlap, count, tuple(origins), tuple(crops), tuple(offsets))


def tile_plan(height, width, tile_size=128, tile_overlap_size=14) -> TilePlan:
    tile, overlap = effective_tile(height, width, tile_size, tile_overlap_size)
    return TilePlan(height, width, tile, overlap, _axis(height, tile, overlap), _axis(width, tile, overlap))


def training_tile_grid(height, width, tiles_height_width):
    """Training-side tiling of a rendered frame (TFRecordsCreator.py:125-133): non-overlapping tiles, remainders dropped.
    Returns (rows, cols, [(y0, y1, x0, x1), ...]) in the reference's loop order (row-major; its "x" indexes height)."""
    rows, cols = height // tiles_height_width, width // tiles_height_width
    t = tiles_height_width
    return rows, cols, [(i * t, (i + 1) * t, j * t, (j + 1) * t) for i in range(rows) for j in range(cols)]


def source_index_tuples(number_of_sources_per_example, number_of_source_index_tuples, number_of_sources_per_target, rng=None):
    """Which of an example's source renderings feed each training tuple (Training.py:879-913).

    One source per target: as many full sweeps 0..S-1 as fit, the remainder drawn with `rng.randint(0, S-1)`; two sources per
    target: distinct pairs by rejection.  `rng` defaults to Python's global `random` module, which is what the reference draws
    from, so a seeded run reproduces its tuples draw for draw.  Returns (index_tuples, sorted unique indices)."""
    import random as _random
    rng = _random if rng is None else rng
    S, n, per = number_of_sources_per_example, number_of_source_index_tuples, number_of_sources_per_target
    if S < per:
        raise Exception("The source index tuples contain unique indices. That is not possible if there are fewer source examples "
                        "than indices per tuple.")
    if per == 1:
        tuples = [[i] for _ in range(n // S) for i in range(S)]
        tuples += [[rng.randint(0, S - 1)] for _ in range(n % S)]
    else:
        if per > 2:
            raise Exception("More than two source inputs are currently not supported!")
        tuples = []
        for _ in range(n):
            picked = []
            while len(picked) < per:
                i = rng.randint(0, S - 1)
                if i not in picked:
                    picked.append(i)
            tuples.append(picked)
    return tuples, sorted({i for t in tuples for i in t})
