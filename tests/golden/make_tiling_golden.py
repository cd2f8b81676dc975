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


def main():
    ref = load_reference()
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
    with open(os.path.join(HERE, "tiling_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_tiling_golden.py",
                   "reference": "qurator/sbb_textline_detector/main.py:225-366 (imported, stubbed deps)",
                   "cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
