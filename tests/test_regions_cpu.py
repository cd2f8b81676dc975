"""Closed forms behind the owned-region decoder launches (csrc/region.h, round 6) -- no GPU.

The fused page paths launch the decoder of every tile only where the page stitch keeps its labels (plus the halo the later levels
read).  That is only exact if (1) the per-axis OWNED range the library derives equals what the reference's crop + paste loop leaves of
each tile (main.py:276-281, 294-364: oracle/tiling.py restates it, pinned by tests/golden/tiling_golden.json) and (2) the rows a level
is asked to produce cover every tap of every level above.  Both are checked here against brute force."""
import numpy as np
import pytest

from oracle import tiling
from sbb_textline_detection_amd import _capi


def _paste_owner(extent, tile, margin):
    """1-D restatement of the paste loop: owner[q] = index of the LAST tile whose cropped range covers q."""
    tiles = tiling.axis_tiles(extent, tile, margin)
    own = np.full(extent, -1, np.int64)
    for (d, lo, hi, t) in tiles:
        own[d + lo:d + hi] = t
    return tiles, own


EXTENTS_448 = sorted(set([448, 449, 450, 500, 535, 536, 537, 720, 777, 800, 807, 808, 809, 1000, 1080, 1234, 1439, 1440, 1441, 1527, 1528, 1529,
                          2500, 3000, 3500, 4000, 4200] + list(range(448, 1900, 37))))


@pytest.mark.parametrize("tile,margin,extents", [(448, 44, EXTENTS_448), (224, 22, list(range(224, 1100, 13))), (320, 48, list(range(320, 1500, 29))),
                                                 (480, 48, list(range(480, 2000, 41)))])
def test_owned_range_equals_the_reference_paste(tile, margin, extents):
    """(320, 48) / (480, 48): the 320 x 480 model of the fixture -- the margin comes from the WIDTH for both axes (main.py:233)."""
    for extent in extents:
        tiles, own = _paste_owner(extent, tile, margin)
        assert (own >= 0).all()
        n = len(tiles)
        for (d, _lo, _hi, t) in tiles:
            lo, hi = _capi.owned_range(extent, tile, margin, n, t)
            q = np.nonzero(own == t)[0]
            if q.size == 0:
                assert lo == hi, (extent, t)                      # a repeated clamped tile owns nothing: the later copy wins
            else:
                assert (q[0] - d, q[-1] + 1 - d) == (lo, hi) and q.size == hi - lo, (extent, t, lo, hi)
        # the fused paths drop the repeated last tile of an axis (sbbseg_set_dedupe): the shorter grid owns the same pixels
        if n >= 2 and tiles[-1][0] == tiles[-2][0]:
            own2 = np.full(extent, -1, np.int64)
            for t in range(n - 1):
                lo, hi = _capi.owned_range(extent, tile, margin, n - 1, t)
                d = tiles[t][0]
                assert (own2[d + lo:d + hi] == -1).all()
                own2[d + lo:d + hi] = t
            assert (own2 >= 0).all() and np.array_equal(np.minimum(own, n - 2), own2)


def _needed_below(lo, hi, rows_below):
    """rows of the level below that rows [lo, hi) read: a 3x3 conv (zero padded) over the nearest-x2 upsampling."""
    need = set()
    for y in range(lo, hi):
        for dy in (-1, 0, 1):
            u = y + dy
            if 0 <= u < 2 * rows_below:
                need.add(u >> 1)
    return need


@pytest.mark.parametrize("tile,margin", [(448, 44), (224, 22), (320, 48)])
def test_region_rows_cover_every_tap_and_nothing_more(tile, margin):
    sizes = [tile >> k for k in range(5)]
    for extent in [tile, tile + 1, tile + 89, 2 * tile, 2 * tile + 200, 3 * tile + 7, 2500, 3500, 4000]:
        if extent < tile:
            continue
        n = len(tiling.axis_tiles(extent, tile, margin))
        for t in range(n):
            rows = _capi.region_rows(extent, tile, margin, n, t, sizes)
            lo, hi = _capi.owned_range(extent, tile, margin, n, t)
            assert tuple(rows[0]) == (lo, hi)
            for k in range(1, 5):
                plo, phi = int(rows[k - 1][0]), int(rows[k - 1][1])
                need = _needed_below(plo, phi, sizes[k])
                got = set(range(int(rows[k][0]), int(rows[k][1])))
                if not need:
                    assert not got
                else:
                    assert got == set(range(min(need), max(need) + 1)), (extent, t, k, rows[k], min(need), max(need))


def test_baseline_page_keeps_62_percent():
    """The numbers DESIGN.md quotes: a 3500 x 2500 page keeps 8.75 of 14.05 Mpx; an interior tile's rows per level."""
    ys = [_capi.owned_range(3500, 448, 44, 10, t) for t in range(10)]
    xs = [_capi.owned_range(2500, 448, 44, 7, t) for t in range(7)]
    assert sum(h - l for l, h in ys) == 3500 and sum(h - l for l, h in xs) == 2500
    assert ys[8] == (44, 216) and xs[5] == (44, 296)
    rows = _capi.region_rows(3500, 448, 44, 10, 4, [448, 224, 112, 56, 28])
    assert rows.tolist() == [[44, 404], [21, 203], [10, 102], [4, 52], [1, 27]]
