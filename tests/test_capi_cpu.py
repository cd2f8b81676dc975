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
    assert lib.sbbseg_abi_version() == 5


@pytest.mark.parametrize("case", GOLD, ids=lambda c: f"{c['page_h']}x{c['page_w']}_m{c['model_h']}x{c['model_w']}")
def test_tile_grid_equals_reference(lib, case):
    xy, nx, ny = _capi.tile_grid(case["page_h"], case["page_w"], case["model_h"], case["model_w"])
    assert nx * ny == case["n_calls"]
    assert xy.tolist() == [list(c) for c in case["calls_xy"]]


def test_small_page_is_an_error_not_a_crash(lib):
    with pytest.raises(RuntimeError, match="smaller than the model"):
        _capi.tile_grid(300, 500, 448, 448)


def test_cpp_exception_becomes_status_not_abort(lib):
    """No-abort guarantee (sbbseg.h conventions; the reference's callers rely on ordinary exceptions,
    main.py:2061-2157): a std::bad_alloc thrown inside the library -- injected at its next host-allocation
    checkpoint -- comes back as a non-zero status with a message; the process and the library stay usable."""
    assert lib.sbbseg_debug_inject_alloc_failure(1) == 0
    with pytest.raises(RuntimeError, match="out of host memory"):
        _capi.tile_grid(1000, 900, 448, 448)
    xy, nx, ny = _capi.tile_grid(1000, 900, 448, 448)          # the hook disarmed itself; the call works again
    assert (nx, ny) == (3, 3) and xy.shape == (9, 2)
    assert lib.sbbseg_debug_inject_alloc_failure(-1) != 0       # bad argument: status, not a crash
    assert lib.sbbseg_debug_inject_alloc_failure(0) == 0


def test_nearest_map_rule(lib):
    """The library's INTER_NEAREST index rule (page rescale fused into the tile gather) against
    hand-computed cases (odd ratios, up and down) and against the oracle's restatement on the sizes the
    reference's own 2800-rule / x1.2-rule produce (main.py:201-207, fixtures from the imported reference)."""
    from oracle import tiling
    hand = {(3, 7): [0, 0, 0, 1, 1, 2, 2], (5, 3): [0, 1, 3], (4, 6): [0, 0, 1, 2, 2, 3], (2, 5): [0, 0, 0, 1, 1],
            (7, 2): [0, 3], (1, 4): [0, 0, 0, 0]}
    for (src, dst), want in hand.items():
        assert _capi.nearest_map(src, dst).tolist() == want, (src, dst)
    scales = json.load(open(os.path.join(ROOT, "tests", "golden", "tiling_golden.json")))["scale_cases"]
    pairs = [(c["h"], c["img_hight_int"]) for c in scales] + [(c["w"], c["img_width_int"]) for c in scales]
    pairs += [(448, 3500), (3500, 448), (4200, 448), (448, 2800), (611, 1234), (1017, 503)]
    for src, dst in pairs:
        ref = tiling.resize_nearest(np.arange(src).reshape(src, 1), dst, 1)[:, 0]
        got = _capi.nearest_map(src, dst)
        assert np.array_equal(got, ref), (src, dst)
        assert got[0] == 0 and got.max() <= src - 1 and np.all(np.diff(got) >= 0)
    with pytest.raises(RuntimeError):
        _capi.nearest_map(0, 5)


def test_documents_name_only_entry_points_that_exist(lib):
    """INTEGRATION.md / DESIGN.md / README.md are what a maintainer of the reference binds from: every `sbbseg_*` function they name
    (wildcards like `sbbseg_comm_*` and `[_dev]` suffix notation aside) must be an export of the library."""
    exports = set(_capi.EXPORTS)
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for name in sorted(set(re.findall(r"`(sbbseg_[a-z0-9_]+)(?:\[_dev\])?`", text))):
            if name.endswith("_") or name in ("sbbseg_ctx", "sbbseg_run_info", "sbbseg_conv_desc", "sbbseg_h"):
                continue
            candidates = {name, name + "_dev"}
            assert candidates & exports or any(e.startswith(name) for e in exports), f"{doc} names {name}, which libsbbseg does not export"


def test_shipped_library_has_no_wrong_answer_probes(lib):
    """The timing probes that make kernels compute WRONG results on purpose (gathers forced into L2, one weight block, dropped stores:
    SBBSEG_CONV_PROBE_LOCAL / _WHOT, SBBSEG_BLOCK_DBG, SBBSEG_ER_DBG) compile only under -DSBBSEG_PROBES (`_build --probes`, a separate
    library under tools/probes/bin/): the library a maintainer binds must not even contain the names of their switches."""
    blob = open(_capi.LIB_PATH, "rb").read()
    for name in (b"SBBSEG_CONV_PROBE_LOCAL", b"SBBSEG_CONV_PROBE_WHOT", b"SBBSEG_BLOCK_DBG", b"SBBSEG_ER_DBG", b"SBBSEG_PROBES"):
        assert name not in blob, name.decode()
    for src in ("api.hip", "kernels.hip", "block_x3.hip", "expand_reduce_x3.hip"):
        text = open(os.path.join(ROOT, "sbb_textline_detection_amd", "csrc", src)).read()
        # every use of the probe bits in device code goes through SBBSEG_PROBE(...) (constant false in the shipped build) ...
        for m in re.finditer(r"(variant_flags & (?:32|64)|p\.dbg & \d)", text):
            line = text[text.rfind("\n", 0, m.start()) + 1:text.find("\n", m.end())]
            assert "SBBSEG_PROBE(" in line or line.lstrip().startswith("//"), (src, line.strip())
        # ... and every getenv of a probe switch sits inside an #ifdef SBBSEG_PROBES block
        depth = 0
        for line in text.splitlines():
            if line.strip().startswith("#ifdef SBBSEG_PROBES"):
                depth += 1
            elif line.strip().startswith("#endif") and depth:
                depth -= 1
            if re.search(r'getenv\("SBBSEG_(CONV_PROBE|BLOCK_DBG|ER_DBG)', line):
                assert depth > 0, (src, line.strip())
