"""oracle/tiling.py (numpy restatement) vs fixtures captured from the imported reference
do_prediction (tests/golden/make_tiling_golden.py; main.py:225-366).  Bit-exact."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import tiling

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))
CASES = GOLD["cases"]
SMALL = [c for c in CASES if c["page_h"] * c["page_w"] <= 1300 * 1300]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_tile_grid_matches_reference_calls(case):
    tiles, nxf, nyf = tiling.tile_grid(case["page_h"], case["page_w"], case["model_h"], case["model_w"])
    assert len(tiles) == case["n_calls"] == nxf * nyf
    assert [[t["x0"], t["y0"]] for t in tiles] == [list(c) for c in case["calls_xy"]]


@pytest.mark.parametrize("case", SMALL + [c for c in CASES if (c["page_h"], c["page_w"]) == (3500, 2500)],
                         ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_do_prediction_matches_reference_output(case):
    page = tiling.coord_page(case["page_h"], case["page_w"])
    fm = tiling.FakeModel(case["model_h"], case["model_w"], case["classes"])
    res = tiling.do_prediction(True, page, fm)
    assert str(res.dtype) == case["out_dtype"] and list(res.shape) == case["out_shape"]
    assert fm.in_dtype == case["predict_in_dtype"] and list(fm.in_shape) == case["predict_in_shape"]
    assert np.array_equal(res[:, :, 0], res[:, :, 1]) and np.array_equal(res[:, :, 0], res[:, :, 2])
    for y, x, v in case["probe"]:
        assert int(res[y, x, 0]) == v
    assert int(res[:, :, 0].astype(np.int64).sum()) == case["out_sum"]
    assert zlib.crc32(np.ascontiguousarray(res[:, :, 0]).tobytes()) & 0xFFFFFFFF == case["out_crc32"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['page_h']}x{c['page_w']}")
def test_owner_map_covers_page_and_is_last_writer(case):
    own = tiling.owner_map(case["page_h"], case["page_w"], case["model_h"], case["model_w"])
    assert own.min() >= 0 and own.max() == case["n_calls"] - 1


def test_small_page_is_unsupported_like_reference():
    # main.py:278/281: page smaller than the model -> negative slice start -> reshape error in the reference
    fm = tiling.FakeModel(448, 448, 4)
    with pytest.raises(Exception):
        tiling.do_prediction(True, tiling.coord_page(300, 500), fm)


def test_resize_nearest_rule():
    a = np.arange(5 * 7).reshape(5, 7)
    up = tiling.resize_nearest(a, 10, 14)
    assert up.shape == (10, 14) and np.array_equal(up[::2, ::2], a)
    dn = tiling.resize_nearest(a, 2, 3)
    assert np.array_equal(dn, a[[0, 2]][:, [0, 2, 4]])


# ---- whole-image branch (patches=False, main.py:368-380) and the page rescale (main.py:196-214): fixtures from the
# IMPORTED reference run with a stub cv2.resize (the restated index rule) -- they pin what is resized to which size in
# which order (incl. the `self.image.shape` quirk of main.py:378); the cv2 index rule itself stays [EXT]
@pytest.mark.parametrize("case", GOLD["whole_cases"], ids=lambda c: f"{c['page_h']}x{c['page_w']}_to_{c['full_h']}x{c['full_w']}")
def test_whole_image_branch_matches_reference_output(case):
    page = tiling.coord_page(case["page_h"], case["page_w"])
    fm = tiling.FakeModel(case["model_h"], case["model_w"], case["classes"])
    res = tiling.do_prediction(False, page, fm, full_image_shape=(case["full_h"], case["full_w"], 3))
    assert str(res.dtype) == case["out_dtype"] and list(res.shape) == case["out_shape"]
    assert fm.in_dtype == case["predict_in_dtype"] and list(fm.in_shape) == case["predict_in_shape"]
    assert int(res.astype(np.int64).sum()) == case["out_sum"]
    assert zlib.crc32(np.ascontiguousarray(res).tobytes()) & 0xFFFFFFFF == case["out_crc32"]


def test_resize_nearest_hand_computed_cases():
    """cv2.INTER_NEAREST index rule [EXT: OpenCV resizeNN, src = min(floor(dst * (1/(dst_len/src_len))), src_len-1)],
    worked by hand for odd ratios (not centre-aligned: index 0 always maps to 0, the last output may not reach the last input)."""
    hand = {(3, 7): [0, 0, 0, 1, 1, 2, 2],          # 1/(7/3) = 0.428571...: 0 .43 .86 1.29 1.71 2.14 2.57
            (5, 3): [0, 1, 3],                      # 1/(3/5) = 1.6667: 0 1.67 3.33
            (4, 6): [0, 0, 1, 2, 2, 3],             # 0.6667: 0 .67 1.33 2 2.67 3.33
            (2, 5): [0, 0, 0, 1, 1],                # 0.4: 0 .4 .8 1.2 1.6
            (7, 2): [0, 3],                         # 3.5: 0 3.5
            (1, 4): [0, 0, 0, 0]}
    for (src, dst), want in hand.items():
        a = np.arange(src).reshape(src, 1)
        assert tiling.resize_nearest(a, dst, 1)[:, 0].tolist() == want, (src, dst)
        assert tiling.resize_nearest(a.T, 1, dst)[0].tolist() == want, (src, dst)


class _ContentModel:
    """A model whose output depends on the PATCH CONTENT only (as a real network's does) -- unlike FakeModel, whose label also
    depends on the call index."""

    class _L:
        def __init__(self, shp):
            self.output_shape = shp

    def __init__(self, H, W, C=8):
        self.layers = [self._L((None, H, W, C))]
        self.H, self.W, self.C, self.calls = H, W, C, 0

    def predict(self, x):
        self.calls += 1
        p = np.rint(np.asarray(x[0], np.float64) * 255.0).astype(np.int64)
        yy, xx = np.mgrid[0:self.H, 0:self.W]
        lab = (yy * 3 + xx * 7 + p[:, :, 0] + 2 * p[:, :, 1] + 5 * p[:, :, 2] + (p[::-1, ::-1, 0])) % self.C
        out = np.zeros((1, self.H, self.W, self.C), np.float32)
        np.put_along_axis(out[0], lab[:, :, None], 1.0, axis=2)
        return out


@pytest.mark.parametrize("hp,wp,distinct,calls", [(390, 640, 8, 12), (640, 404, 8, 12), (224, 367, 2, 6), (405, 405, 9, 9), (224, 224, 1, 4),
                                                   (225, 224 + 180, 4, 6), (3 * 180 + 1, 5 * 180 + 44, 15, 24)])
def test_repeated_clamped_tiles_can_be_computed_once(hp, wp, distinct, calls):
    """The library's fused page paths drop the reference's repeated forward when the clamp (main.py:276-281) gives the last two
    tiles of an axis one origin (sbbseg_set_dedupe).  Claim checked here on the CPU, against the restated reference loop: pasting
    the grid WITHOUT the repeated tile -- the remaining last tile pasting through to the edge -- gives the same map, for any model
    that is a function of the patch content."""
    H = W = 224
    margin = int(0.1 * W)
    rng = np.random.RandomState(hp * 31 + wp)
    page = rng.randint(0, 256, (hp, wp, 3)).astype(np.uint8)
    m = _ContentModel(H, W)
    ref = tiling.do_prediction(True, page, m)[:, :, 0]
    assert m.calls == calls

    def dedup(axis):
        t = tiling.axis_tiles(axis, H, margin)
        if len(t) >= 2 and t[-1][0] == t[-2][0]:
            t = t[:-1]
            d, lo, _, k = t[-1]
            t[-1] = (d, lo, H, k)                              # the new last tile keeps its far side
        return t
    xs, ys = dedup(wp), dedup(hp)
    assert len(xs) * len(ys) == distinct
    out = np.zeros((hp, wp), np.uint8)
    m2 = _ContentModel(H, W)
    img = page / 255.0
    for (x0, xlo, xhi, _) in xs:
        for (y0, ylo, yhi, _) in ys:
            lab = np.argmax(m2.predict(img[y0:y0 + H, x0:x0 + W][None]), axis=3)[0]
            out[y0 + ylo:y0 + yhi, x0 + xlo:x0 + xhi] = lab[ylo:yhi, xlo:xhi]
    assert m2.calls == distinct
    assert np.array_equal(out, ref)
