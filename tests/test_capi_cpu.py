"""C-ABI checks that need no GPU: the library loads, exports every symbol include/sbbseg.h declares,
reports errors as return codes (never aborts), and its tile grid equals the reference's."""
import json
import os
import re

import numpy as np
import pytest

from sbb_textline_detection_amd import _build, _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "tiling_golden.json")))["cases"]


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return _capi.load_library()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "sbbseg.h")).read()
    declared = sorted(set(re.findall(r"\b(sbbseg_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sbbseg.h but not exported"
    assert sorted(_capi.EXPORTS) == declared


def test_abi_version(lib):
    assert lib.sbbseg_abi_version() == 2


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_tile_grid_equals_reference(lib, case):
    xy, nx, ny = _capi.tile_grid(case["page_h"], case["page_w"], case["model_h"], case["model_w"])
    assert nx * ny == case["n_calls"]
    assert xy.tolist() == [list(c) for c in case["calls_xy"]]


def test_small_page_is_an_error_not_a_crash(lib):
    with pytest.raises(RuntimeError, match="smaller than the model"):
        _capi.tile_grid(300, 500, 448, 448)
