"""oracle/stage_glue.py (CPU): the Otsu restatement against hand-derived cases, a brute-force
between-class-variance maximiser, and the host mirror in stages.py.  The cv2 arithmetic itself is
unpinned (no OpenCV in this image, no vectors in the reference) -- see the oracle's header."""
import numpy as np

from oracle import stage_glue
from sbb_textline_detection_amd import stages
from sbb_textline_detection_amd.synthetic import synthetic_page


def _hist(pairs):
    h = np.zeros(256, np.int64)
    for v, n in pairs:
        h[v] = n
    return h


def test_two_spikes_first_maximum_wins():
    # two equal spikes at 50 and 200: sigma is the same for every t in [50, 199]; cv2's strict `>` keeps the first
    assert stage_glue.otsu_threshold_from_hist(_hist([(50, 1000), (200, 1000)])) == 50
    assert stage_glue.otsu_threshold_from_hist(_hist([(0, 1), (255, 1)])) == 0
    # unequal spikes: still any t in [50, 199] separates them perfectly -> first
    assert stage_glue.otsu_threshold_from_hist(_hist([(50, 10), (200, 990)])) == 50


def test_constant_page_gives_zero():
    # one bin holds everything: q1/q2 test skips every i -> max_val stays 0 (dst = src > 0)
    assert stage_glue.otsu_threshold_from_hist(_hist([(255, 12345)])) == 0
    assert stage_glue.otsu_threshold_from_hist(_hist([(0, 7)])) == 0


def test_three_level_case_by_hand():
    # values 10 (x3), 20 (x1), 110 (x4): candidates t=10: q1=3/8, mu1=10, mu2=(20+440)/5=92 -> 0.375*0.625*82^2 = 1575.9
    #                                   t=20..109: q1=.5, mu1=12.5, mu2=110 -> .25*97.5^2 = 2376.6  -> first such t = 20
    assert stage_glue.otsu_threshold_from_hist(_hist([(10, 3), (20, 1), (110, 4)])) == 20


def _bruteforce(hist):
    p = hist.astype(np.float64) / hist.sum()
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


def test_against_bruteforce_and_host_mirror():
    rng = np.random.RandomState(0)
    for seed in range(6):
        page = synthetic_page(300 + 40 * seed, 280, seed=seed)
        for c in range(3):
            t = stage_glue.otsu_threshold(page[:, :, c])
            assert abs(t - _bruteforce(stage_glue.histogram_u8(page[:, :, c]))) <= 1
            assert t == stages.otsu_threshold_u8(page[:, :, c])          # product host mirror, same arithmetic
    for _ in range(20):                                                  # random smooth-ish histograms
        h = (rng.gamma(0.6, 200.0, 256) * (rng.rand(256) < 0.7)).astype(np.int64)
        h[rng.randint(256)] += 5
        t = stage_glue.otsu_threshold_from_hist(h)
        assert abs(t - _bruteforce(h)) <= 1


def test_otsu_copy_quirk():
    page = synthetic_page(200, 160, seed=4)
    page[:, :, 1] = 255 - page[:, :, 1]                                  # channels differ: only channel 0 may matter
    out = stage_glue.otsu_copy(page)
    assert out.dtype == np.float64 and out.shape == page.shape
    t = stage_glue.otsu_threshold(page[:, :, 0])
    for c in range(3):                                                   # main.py:191-193
        assert np.array_equal(out[:, :, c], np.where(page[:, :, 0] > t, 255.0, 0.0))
    assert np.array_equal(out, stages.otsu_copy(page))


# ---- morphology / page box (SURVEY 8f-3 remainder)
def test_morph_hand_cases():
    a = np.zeros((9, 11), np.uint8)
    a[4, 5] = 255
    d = stage_glue.morph(a, "dilate", 5, 1)
    assert d.sum() == 25 * 255 and d[2:7, 3:8].min() == 255                  # one pixel -> 5x5 block
    d2 = stage_glue.morph(a, "dilate", 5, 2)
    assert np.count_nonzero(d2) == 9 * 9                                      # two iterations == one 9x9 (clipped: fits here)
    assert np.array_equal(stage_glue.morph(d, "erode", 5, 1), a)              # erosion of the block gives the pixel back
    # border: the outside never wins -> a full plane stays full under erosion, an empty one stays empty under dilation
    assert stage_glue.morph(np.full((6, 7), 255, np.uint8), "erode", 5, 3).min() == 255
    assert stage_glue.morph(np.zeros((6, 7), np.uint8), "dilate", 5, 6).max() == 0
    # grey levels (the layout map holds classes 0..3): min / max, not binary logic
    g = np.array([[3, 3, 3, 3, 3, 3, 3], [3, 1, 3, 3, 3, 2, 3]], np.uint8)
    assert stage_glue.morph(g, "erode", 5, 1).tolist() == [[1, 1, 1, 1, 2, 2, 2]] * 2
    # corner pixel: clipped window
    c = np.zeros((8, 8), np.uint8); c[0, 0] = 7
    assert np.count_nonzero(stage_glue.morph(c, "dilate", 5, 1)) == 9


def test_host_mirrors_equal_oracle_morph_and_box():
    rng = np.random.RandomState(5)
    for shape in ((40, 57), (64, 64), (33, 90)):
        m = (rng.rand(*shape) < 0.08).astype(np.uint8) * rng.randint(1, 4, shape).astype(np.uint8)
        for op, it in (("erode", 3), ("dilate", 4), ("dilate", 6), ("erode", 1)):
            assert np.array_equal(stages.host_morph(m, op == "dilate", 5, it), stage_glue.morph(m, op, 5, it)), (shape, op, it)
        assert stages.host_page_box(m) == stage_glue.page_box(m)
    assert stage_glue.page_box(np.zeros((30, 30), np.uint8)) == ((0, 0, 0, 0), 0)


def test_page_box_by_hand():
    m = np.zeros((100, 120), np.uint8)
    m[30:60, 40:90] = 1                      # the page blob: 30 x 50
    m[5:8, 5:8] = 1                          # a speck far away (more than 24 px: stays a separate component after dilation)
    box, px = stage_glue.page_box(np.repeat(m[:, :, None], 3, axis=2))
    assert box == (28, 18, 74, 54)           # dilated by 12 on every side: x 40-12 .. 89+12, y 30-12 .. 59+12
    assert px == 74 * 54
    crop, coord = stage_glue.crop_image_inside_box(box, np.zeros((100, 120, 3), np.uint8))
    assert crop.shape == (54, 74, 3) and coord == [18, 72, 28, 102]          # main.py:174-176: [y, y+h, x, x+w]
    # 8-connectivity: two blobs touching at a corner are ONE component
    d = np.zeros((40, 40), np.uint8); d[0:10, 0:10] = 1; d[10:20, 10:20] = 1
    assert stage_glue.largest_component_box(d) == ((0, 0, 20, 20), 200)


def test_morph_and_components_against_scipy():
    """An implementation that shares no code with the oracle's: scipy.ndimage's grey erosion / dilation with a 5 x 5 flat footprint and a
    constant border that can never win (cv2's morphologyDefaultBorderValue), iterated; and ndimage.label (8-connectivity) +
    find_objects for the bounding box of the largest blob where one blob is clearly the largest (no area ties, the one case where
    cv2.contourArea's ranking and a pixel count can differ is avoided by construction: solid rectangles)."""
    from scipy import ndimage
    rng = np.random.RandomState(11)
    for shape in ((37, 53), (64, 64), (5, 9), (120, 31)):
        grey = rng.randint(0, 4, shape).astype(np.uint8)                      # layout classes 0..3
        binary = ((rng.rand(*shape) > 0.8) * 255).astype(np.uint8)
        for a in (grey, binary):
            for it in (1, 3, 4, 6):
                e, d = a.copy(), a.copy()
                for _ in range(it):
                    e = ndimage.grey_erosion(e, size=(5, 5), mode="constant", cval=255)
                    d = ndimage.grey_dilation(d, size=(5, 5), mode="constant", cval=0)
                assert np.array_equal(stage_glue.morph(a, "erode", 5, it), e), (shape, it)
                assert np.array_equal(stage_glue.morph(a, "dilate", 5, it), d), (shape, it)
    m = np.zeros((90, 70), np.uint8)
    m[10:50, 5:40] = 1                                                      # 40 x 35, the largest
    m[60:80, 30:65] = 1                                                     # 20 x 35
    m[2:6, 60:68] = 1
    lab, n = ndimage.label(m, structure=np.ones((3, 3), int))
    sizes = ndimage.sum(m, lab, range(1, n + 1))
    sl = ndimage.find_objects(lab)[int(np.argmax(sizes))]
    box, _ = stage_glue.largest_component_box(m)
    assert tuple(box) == (sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start) == (5, 10, 35, 40)
