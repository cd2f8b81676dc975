#!/usr/bin/env python3
"""Generate tests/golden/tiling_golden.json by running the REAL reference patch loop.

Runs only in the build container (needs /root/reference); never on the GPU box.  The reference's
``main.py`` imports cv2 / seaborn / keras / tensorflow / shapely at module top -- none installed --
so empty stub modules are registered first; ``do_prediction(patches=True, ...)`` itself only
needs numpy and a model duck type (main.py:225-366).  Nothing from the reference is copied:
the fixture holds inputs' descriptions (page sizes) and expected outputs (tile origins, CRCs).
"""
import importlib.util
import json
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.tiling import FakeModel, coord_page  # noqa: E402  (test drivers only)

REF_MAIN = "/root/reference/qurator/sbb_textline_detector/main.py"

# (page_h, page_w, model_h, model_w)
CASES = [
    (448, 448, 448, 448),      # 4 identical clamped tiles
    (449, 1000, 448, 448),     # 1-px overhang
    (800, 800, 448, 448),
    (720, 1080, 448, 448),     # exact multiples of mid=360
    (777, 1234, 448, 448),
    (1234, 777, 448, 448),
    (808, 720, 448, 448),
    (3500, 2500, 448, 448),    # BASELINE config[1]
    (4000, 3000, 448, 448),    # BASELINE config[3]
    (4200, 3000, 448, 448),    # config[1] after get_image_and_scales (main.py:205-207)
    (700, 900, 320, 480),      # non-square model: margin derives from width (main.py:233)
    (512, 640, 224, 224),
]


def load_reference():
    for name in ("cv2", "seaborn", "keras", "keras.models", "keras.backend", "tensorflow",
                 "shapely", "shapely.geometry", "matplotlib", "matplotlib.pyplot", "tqdm",
                 "sklearn", "sklearn.cluster", "click"):
        if name not in sys.modules:
            try:
                __import__(name)
                continue
            except Exception:
                pass
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["keras.models"].model_from_json = None
    sys.modules["keras.models"].load_model = None
    sys.modules["keras"].models = sys.modules["keras.models"]
    sys.modules["keras"].backend = sys.modules["keras.backend"]
    sys.modules["shapely"].geometry = sys.modules["shapely.geometry"]
    tf = sys.modules["tensorflow"]
    if not hasattr(tf, "get_logger"):
        tf.get_logger = lambda: types.SimpleNamespace(setLevel=lambda *_: None)
    if not hasattr(sys.modules["tqdm"], "tqdm"):
        sys.modules["tqdm"].tqdm = lambda x, **k: x
    if not hasattr(sys.modules["sklearn.cluster"], "KMeans"):
        sys.modules["sklearn.cluster"].KMeans = None
    spec = importlib.util.spec_from_file_location("ref_main", REF_MAIN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# whole-image branch (patches=False, main.py:368-380): (page_h, page_w, model_h, model_w, full_h, full_w);
# full_* = self.image.shape, which the reference resizes back to (main.py:378) -- NOT img.shape
WHOLE_CASES = [
    (700, 520, 224, 224, 840, 624),
    (448, 448, 448, 448, 448, 448),
    (1234, 777, 448, 448, 1234, 777),
    (611, 503, 320, 480, 2800, 2305),
    (3500, 2500, 448, 448, 4200, 3000),
]
# get_image_and_scales (main.py:196-214): stored (h, w) -> the size the page is upscaled to
SCALE_CASES = [(520, 400), (900, 700), (2499, 1800), (2500, 1800), (3500, 2500), (2501, 3333), (1000, 1000), (4000, 3000), (3508, 2481)]


def stub_cv2_resize(img, dsize, interpolation=None, **_kw):
    """Stand-in for cv2.resize(..., INTER_NEAREST) inside the IMPORTED reference: the restated OpenCV resizeNN index
    rule [EXT: src = min(floor(dst * (1 / (dst_len / src_len))), src_len - 1)].  It pins the reference's branch
    STRUCTURE (what is resized to which size, in which order, with which dtypes) -- not the index rule itself."""
    out_w, out_h = int(dsize[0]), int(dsize[1])
    in_h, in_w = img.shape[:2]
    xs = [min(int(np.floor(i * (1.0 / (out_w / float(in_w))))), in_w - 1) for i in range(out_w)]
    ys = [min(int(np.floor(i * (1.0 / (out_h / float(in_h))))), in_h - 1) for i in range(out_h)]
    return img[np.asarray(ys)][:, np.asarray(xs)]


def main():
    ref = load_reference()
    import cv2 as cv2_stub
    cv2_stub.resize = stub_cv2_resize
    cv2_stub.INTER_NEAREST = 0
    det = ref.textline_detector.__new__(ref.textline_detector)
    out = []
    for (ph, pw, mh, mw) in CASES:
        page = coord_page(ph, pw)
        fm = FakeModel(mh, mw, 16)
        res = det.do_prediction(True, page, fm)
        assert res.dtype == np.uint8 and res.shape == (ph, pw, 3)
        assert np.array_equal(res[:, :, 0], res[:, :, 1]) and np.array_equal(res[:, :, 0], res[:, :, 2])
        out.append({
            "page_h": ph, "page_w": pw, "model_h": mh, "model_w": mw, "classes": 16,
            "n_calls": len(fm.calls), "calls_xy": fm.calls,
            "predict_in_dtype": fm.in_dtype, "predict_in_shape": list(fm.in_shape),
            "out_dtype": str(res.dtype), "out_shape": list(res.shape),
            "out_crc32": zlib.crc32(np.ascontiguousarray(res[:, :, 0]).tobytes()) & 0xFFFFFFFF,
            "out_sum": int(res[:, :, 0].astype(np.int64).sum()),
            "probe": [[int(y), int(x), int(res[y, x, 0])] for (y, x) in
                      [(0, 0), (ph - 1, pw - 1), (ph // 2, pw // 2), (ph - 1, 0), (0, pw - 1),
                       (min(403, ph - 1), min(404, pw - 1)), (min(404, ph - 1), min(403, pw - 1))]],
        })
        print(ph, pw, mh, mw, "calls", len(fm.calls), "crc", out[-1]["out_crc32"])
    whole = []
    for (ph, pw, mh, mw, fh, fw) in WHOLE_CASES:
        page = coord_page(ph, pw)
        fm = FakeModel(mh, mw, 16)
        det.image = np.zeros((fh, fw, 3), np.uint8)                     # self.image: only its shape is read (main.py:378)
        res = det.do_prediction(False, page, fm)
        assert res.dtype == np.uint8 and res.shape == (fh, fw, 3)
        whole.append({
            "page_h": ph, "page_w": pw, "model_h": mh, "model_w": mw, "full_h": fh, "full_w": fw, "classes": 16,
            "predict_in_dtype": fm.in_dtype, "predict_in_shape": list(fm.in_shape),
            "out_dtype": str(res.dtype), "out_shape": list(res.shape),
            "out_crc32": zlib.crc32(np.ascontiguousarray(res).tobytes()) & 0xFFFFFFFF,
            "out_sum": int(res.astype(np.int64).sum())})
        print("whole", ph, pw, mh, mw, fh, fw, "crc", whole[-1]["out_crc32"])
    scales = []
    for (h, w) in SCALE_CASES:
        cv2_stub.imread = lambda _p, h=h, w=w: np.zeros((h, w, 3), np.uint8)
        det.image_dir = "unused"
        det.get_image_and_scales()
        assert det.image.shape == (det.img_hight_int, det.img_width_int, 3)
        scales.append({"h": h, "w": w, "img_hight_int": int(det.img_hight_int), "img_width_int": int(det.img_width_int),
                       "scale_y": float(det.scale_y), "scale_x": float(det.scale_x)})
        print("scale", h, w, "->", det.img_hight_int, det.img_width_int)
    with open(os.path.join(HERE, "tiling_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_tiling_golden.py",
                   "reference": "qurator/sbb_textline_detector/main.py:196-214, 225-380 (imported, stubbed deps)",
                   "cv2_resize": "stub implementing the restated OpenCV resizeNN index rule [EXT]: the whole-image and "
                                 "rescale fixtures pin the reference's branch structure, not that rule",
                   "cases": out, "whole_cases": whole, "scale_cases": scales}, f, indent=1)


if __name__ == "__main__":
    main()
