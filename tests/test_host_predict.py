"""predict.do_prediction's host (generic-model) path against the fixtures captured from the
reference loop, plus the weight container and model-path plumbing.  CPU only."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import tiling
from sbb_textline_detection_amd import predict
from sbb_textline_detection_amd.model import resolve_model_path
from sbb_textline_detection_amd.weights import load_sbbw, save_sbbw, synthetic_model

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["cases"]
SMALL = [c for c in GOLD if c["page_h"] * c["page_w"] <= 1300 * 1300]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_host_do_prediction_matches_reference(case):
    page = tiling.coord_page(case["page_h"], case["page_w"])
    fm = tiling.FakeModel(case["model_h"], case["model_w"], case["classes"])
    res = predict.do_prediction(True, page, fm)
    assert res.dtype == np.uint8 and list(res.shape) == case["out_shape"]
    assert [list(c) for c in fm.calls] == [list(c) for c in case["calls_xy"]]
    assert fm.in_dtype == case["predict_in_dtype"] and list(fm.in_shape) == case["predict_in_shape"]
    assert zlib.crc32(np.ascontiguousarray(res[:, :, 0]).tobytes()) & 0xFFFFFFFF == case["out_crc32"]
    assert np.array_equal(res[:, :, 0], res[:, :, 2])


def test_whole_image_branch_matches_oracle_restatement():
    page = tiling.coord_page(500, 700)
    fm1, fm2 = tiling.FakeModel(224, 224, 4), tiling.FakeModel(224, 224, 4)
    a = predict.do_prediction(False, page, fm1, full_image_shape=(640, 800, 3))
    b = tiling.do_prediction(False, page, fm2, full_image_shape=(640, 800, 3))
    assert a.shape == (640, 800, 3) and a.dtype == np.uint8 and np.array_equal(a, b)


def test_patch_segmenter_signature():
    seg = predict.PatchSegmenter(image=np.zeros((640, 800, 3), np.uint8))
    fm = tiling.FakeModel(224, 224, 4)
    out = seg.do_prediction(False, tiling.coord_page(500, 700), fm)
    assert out.shape == (640, 800, 3)
    assert seg.resize_image(np.zeros((10, 20, 3)), 5, 8).shape == (5, 8, 3)


def test_small_page_raises():
    with pytest.raises(ValueError):
        predict.do_prediction(True, tiling.coord_page(300, 500), tiling.FakeModel(448, 448, 2))


def test_sbbw_roundtrip(tmp_path):
    cfg, w = synthetic_model(2, 64, 64, seed=4)
    p = str(tmp_path / "model_textline_new.sbbw")
    save_sbbw(p, cfg, w)
    cfg2, w2 = load_sbbw(p)
    assert cfg2 == json.loads(json.dumps(cfg))
    assert set(w2) == set(w) and all(np.array_equal(w[k], w2[k]) for k in w)
    # the reference passes <dir>/model_*.h5 (main.py:58-60): resolve to the converted sibling
    assert resolve_model_path(str(tmp_path / "model_textline_new.h5")) == p
    with pytest.raises(FileNotFoundError):
        resolve_model_path(str(tmp_path / "missing.h5"))


def test_host_whole_image_branch_matches_reference():
    """predict.do_prediction(patches=False) against the fixtures of the imported reference (stub cv2.resize)."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["whole_cases"]
    for case in gold:
        if case["page_h"] * case["page_w"] > 1300 * 1300:
            continue
        page = tiling.coord_page(case["page_h"], case["page_w"])
        fm = tiling.FakeModel(case["model_h"], case["model_w"], case["classes"])
        res = predict.do_prediction(False, page, fm, full_image_shape=(case["full_h"], case["full_w"], 3))
        assert res.dtype == np.uint8 and list(res.shape) == case["out_shape"]
        assert zlib.crc32(np.ascontiguousarray(res).tobytes()) & 0xFFFFFFFF == case["out_crc32"]
