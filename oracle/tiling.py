"""ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's
patch-loop ``textline_detector.do_prediction`` (``/root/reference/qurator/sbb_textline_detector/
main.py:225-380``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Pinned: ``tests/golden/tiling_golden.json`` was produced by importing the *real* reference
``main.py`` (stub modules for cv2/keras/tf, see ``tests/golden/make_tiling_golden.py``) and driving
its ``do_prediction(patches=True)`` with a deterministic fake model; ``tests/test_oracle_tiling.py``
checks this restatement against it bit-exactly.  The whole-image branch (``patches=False``)
depends on ``cv2.resize(INTER_NEAREST)`` which is not installed here -> its index rule is restated
from OpenCV 4.5.1's resizeNN [EXT, unpinned].

Each step cites the reference line it follows.
"""
from __future__ import annotations

import math
from typing import Callable, List, Tuple

import numpy as np


def model_hwc(model) -> Tuple[int, int, int]:
    """main.py:227-229 -- H, W, C from the last layer's output_shape."""
    shp = model.layers[len(model.layers) - 1].output_shape
    return int(shp[1]), int(shp[2]), int(shp[3])


def axis_tiles(extent: int, tile: int, margin: int) -> List[Tuple[int, int, int, int]]:
    """Per-axis tile list [(origin, crop_lo, crop_hi, index)].

    main.py:233-236  mid = tile - 2*margin
    main.py:246-257  count = ceil(extent / mid)
    main.py:262-281  origin = t*mid, clamped inward so that origin+tile <= extent
    main.py:294-364  crop `margin` on every side that is not the first/last tile of the axis
    """
    mid = tile - 2 * margin
    n = extent / float(mid)
    n = int(n) + 1 if n > int(n) else int(n)
    out = []
    for t in range(n):
        d = t * mid
        u = d + tile
        if u > extent:
            u = extent
            d = extent - tile
        lo = 0 if t == 0 else margin
        hi = tile if t == n - 1 else tile - margin
        out.append((d, lo, hi, t))
    return out


def tile_grid(img_h: int, img_w: int, H: int, W: int):
    """All tiles in the reference's call order (x outer, y inner: main.py:259-260).
    NB the reference derives the margin from the model *width* for both axes (main.py:233)."""
    margin = int(0.1 * W)
    xs = axis_tiles(img_w, W, margin)
    ys = axis_tiles(img_h, H, margin)
    tiles = []
    for (x0, xlo, xhi, i) in xs:
        for (y0, ylo, yhi, j) in ys:
            tiles.append({"i": i, "j": j, "x0": x0, "y0": y0, "xlo": xlo, "xhi": xhi, "ylo": ylo, "yhi": yhi})
    return tiles, len(xs), len(ys)


def owner_map(img_h: int, img_w: int, H: int, W: int) -> np.ndarray:
    """int32 [img_h, img_w]: call index (i*nyf + j) of the tile whose label survives at each pixel
    (later calls overwrite earlier ones: main.py:298-364 are plain slice assignments)."""
    tiles, nxf, nyf = tile_grid(img_h, img_w, H, W)
    own = np.full((img_h, img_w), -1, np.int32)
    for k, t in enumerate(tiles):
        own[t["y0"] + t["ylo"]:t["y0"] + t["yhi"], t["x0"] + t["xlo"]:t["x0"] + t["xhi"]] = k
    return own


def resize_nearest(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_NEAREST) restated (main.py:112-113).
    OpenCV resizeNN [EXT]: fx = dst/src (double); ifx = 1/fx; src = min(floor(dst_idx * ifx), src_len-1)."""
    in_h, in_w = img.shape[:2]
    ifx = 1.0 / (out_w / float(in_w))
    ify = 1.0 / (out_h / float(in_h))
    xs = np.minimum(np.floor(np.arange(out_w) * ifx).astype(np.int64), in_w - 1)
    ys = np.minimum(np.floor(np.arange(out_h) * ify).astype(np.int64), in_h - 1)
    return img[ys][:, xs]


def do_prediction(patches: bool, img: np.ndarray, model, full_image_shape=None) -> np.ndarray:
    """Restatement of main.py:225-380.  ``model`` needs ``.layers[-1].output_shape`` and ``.predict``.
    ``full_image_shape`` stands in for ``self.image.shape`` of the whole-image branch (main.py:378)."""
    H, W, _C = model_hwc(model)
    if patches:
        x = img / float(255.0)                                           # main.py:239 (float64)
        img_h, img_w = x.shape[0], x.shape[1]
        out = np.zeros((img_h, img_w, 3))                                # main.py:244 (float64)
        tiles, _, _ = tile_grid(img_h, img_w, H, W)
        for t in tiles:
            patch = x[t["y0"]:t["y0"] + H, t["x0"]:t["x0"] + W, :]       # main.py:285
            probs = model.predict(patch.reshape(1, patch.shape[0], patch.shape[1], patch.shape[2]))  # 287-288
            seg = np.argmax(probs, axis=3)[0]                            # main.py:290 (first max wins)
            seg = seg[t["ylo"]:t["yhi"], t["xlo"]:t["xhi"]]              # main.py:294-364 crop
            out[t["y0"] + t["ylo"]:t["y0"] + t["yhi"], t["x0"] + t["xlo"]:t["x0"] + t["xhi"], :] = seg[:, :, None]
        return out.astype(np.uint8)                                      # main.py:366
    x = img / float(255.0)                                               # main.py:370
    x = resize_nearest(x, H, W)                                          # main.py:371
    probs = model.predict(x.reshape(1, x.shape[0], x.shape[1], x.shape[2]))   # main.py:373-374
    seg = np.argmax(probs, axis=3)[0]                                    # main.py:376
    seg3 = np.repeat(seg[:, :, np.newaxis], 3, axis=2)                   # main.py:377
    shp = full_image_shape if full_image_shape is not None else img.shape
    return resize_nearest(seg3, shp[0], shp[1]).astype(np.uint8)         # main.py:378-379


def normalise_u8(page_u8: np.ndarray) -> np.ndarray:
    """float32 view of main.py:239: f64 divide then the f32 cast Keras applies at predict()."""
    return (page_u8 / float(255.0)).astype(np.float32)


class FakeModel:
    """Deterministic stand-in for the Keras model used to pin the tiling (no network involved).
    The page is expected to encode coordinates (see :func:`coord_page`), which lets predict()
    record each call's tile origin; the label depends on call index, in-patch position and content."""

    class _L:
        def __init__(self, shp):
            self.output_shape = shp

    def __init__(self, H=448, W=448, C=16):
        self.layers = [self._L((None, H, W, C))]
        self.H, self.W, self.C = H, W, C
        self.calls = []
        self.in_dtype = None
        self.in_shape = None

    def predict(self, x):
        self.in_dtype, self.in_shape = str(x.dtype), tuple(x.shape)
        p = np.rint(np.asarray(x[0], np.float64) * 255.0).astype(np.int64)
        x0 = int(p[0, 0, 0] + 256 * (p[0, 0, 2] // 16))
        y0 = int(p[0, 0, 1] + 256 * (p[0, 0, 2] % 16))
        k = len(self.calls)
        self.calls.append((x0, y0))
        yy, xx = np.mgrid[0:self.H, 0:self.W]
        lab = (k * 5 + yy * 3 + xx * 7 + p[:, :, 0] + 2 * p[:, :, 1]) % self.C
        out = np.zeros((1, self.H, self.W, self.C), np.float32)
        np.put_along_axis(out[0], lab[:, :, None], 1.0, axis=2)
        return out


def coord_page(h: int, w: int) -> np.ndarray:
    """uint8 [h,w,3] page whose pixel values encode their own coordinates (h,w <= 4096)."""
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([xx % 256, yy % 256, (xx // 256) * 16 + (yy // 256)], axis=2).astype(np.uint8)
