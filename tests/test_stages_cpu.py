"""Host-side pieces of the stage wrappers (CPU): upscale rule, Otsu restatement."""
import numpy as np

from sbb_textline_detection_amd import stages
from sbb_textline_detection_amd.synthetic import synthetic_page


def test_scaled_size_rule():
    assert stages.scaled_size(2000, 1500) == (2800, 2100)                 # main.py:201-203
    assert stages.scaled_size(3500, 2500) == (4200, 3000)                 # main.py:205-207 (BASELINE config[1])
    assert stages.scaled_size(2500, 1801) == (3000, int(3000 * 1801 / 2500.0))


def _otsu_bruteforce(ch):
    # textbook between-class variance maximisation, first maximum wins
    hist = np.bincount(ch.reshape(-1), minlength=256).astype(np.float64)
    p = hist / hist.sum()
    best, bs = 0, 0.0
    for t in range(256):
        q1, q2 = p[:t + 1].sum(), p[t + 1:].sum()
        if q1 < 1e-7 or q2 < 1e-7:
            continue
        m1 = (np.arange(t + 1) * p[:t + 1]).sum() / q1
        m2 = (np.arange(t + 1, 256) * p[t + 1:]).sum() / q2
        s = q1 * q2 * (m1 - m2) ** 2
        if s > bs * (1 + 1e-12):
            bs, best = s, t
    return best


def test_otsu_matches_bruteforce_and_quirk():
    page = synthetic_page(400, 300, seed=2)
    for c in range(3):
        assert abs(stages.otsu_threshold_u8(page[:, :, c]) - _otsu_bruteforce(page[:, :, c])) <= 1
    out = stages.otsu_copy(page)
    assert out.dtype == np.float64 and out.shape == page.shape and set(np.unique(out)) <= {0.0, 255.0}
    assert np.array_equal(out[:, :, 0], out[:, :, 1]) and np.array_equal(out[:, :, 0], out[:, :, 2])   # main.py:191-193
    t = stages.otsu_threshold_u8(page[:, :, 0])
    assert np.array_equal(out[:, :, 0] > 0, page[:, :, 0] > t)


def test_scaled_size_equals_imported_reference():
    """stages.scaled_size against fixtures produced by the imported reference's get_image_and_scales (main.py:196-214;
    tests/golden/make_tiling_golden.py): the < 2500 -> 2800 rule, the x1.2 rule, and their int() truncations."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tiling_golden.json")))["scale_cases"]
    assert len(gold) >= 8
    for c in gold:
        assert stages.scaled_size(c["h"], c["w"]) == (c["img_hight_int"], c["img_width_int"]), c
        st = stages.InferenceStages("a", "b", "c")
        st.get_image_and_scales(np.zeros((c["h"], c["w"], 3), np.uint8))
        assert (st.scale_y, st.scale_x) == (c["scale_y"], c["scale_x"])
