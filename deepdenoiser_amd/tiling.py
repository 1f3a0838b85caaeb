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
