"""extract_page's blob ranking (main.py:398-404: cv2.findContours + cv2.contourArea + np.argmax + cv2.boundingRect).

cv2 is not installable, so the OpenCV semantics are restated [EXT]: the outer contour runs through the centres of the
component's boundary pixels; contourArea is the shoelace area of that chain.  Three independent statements of that number are
held to each other here: the oracle's border following (oracle/stage_glue.outer_contour_area2, Python), the closed form
(2x2 cells fully inside + half the three-pixel cells, over the hole-filled component -- scipy's binary_fill_holes), and the
library's host tracer (C++, sbbseg_debug_largest_contour).  The -m gpu part checks the device ranking (lower / upper bounds,
host fallback when undecided) against the oracle on hole-bearing masks."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import stage_glue as sg
from sbb_textline_detection_amd import _capi


def closed_form_area2(comp: np.ndarray) -> int:
    f = ndimage.binary_fill_holes(comp, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])).astype(np.int64)      # holes = 4-connected background
    k = f[:-1, :-1] + f[:-1, 1:] + f[1:, :-1] + f[1:, 1:]
    return int(2 * (k == 4).sum() + (k == 3).sum())


def blobs(seed, h=40, w=56, p=0.58):
    rng = np.random.RandomState(seed)
    m = ndimage.binary_opening(rng.rand(h, w) < p, iterations=1)
    m |= rng.rand(h, w) < 0.02                      # specks and diagonal links
    return m


def ring(h, w, t):
    m = np.zeros((h, w), bool)
    m[:t, :] = m[-t:, :] = True
    m[:, :t] = m[:, -t:] = True
    return m


@pytest.mark.parametrize("seed", range(12))
def test_border_following_equals_closed_form(seed):
    m = blobs(seed)
    lab, n = ndimage.label(m, structure=np.ones((3, 3), int))
    assert n > 3
    for k in range(1, n + 1):
        comp = lab == k
        assert sg.outer_contour_area2(comp) == closed_form_area2(comp), (seed, k)


def test_known_areas():
    assert sg.outer_contour_area2(np.ones((1, 1), bool)) == 0
    assert sg.outer_contour_area2(np.ones((1, 9), bool)) == 0                          # a line has no area
    assert sg.outer_contour_area2(np.ones((5, 8), bool)) == 2 * 4 * 7                  # (h - 1)(w - 1)
    assert sg.outer_contour_area2(ring(9, 12, 1)) == 2 * 8 * 11                        # a frame counts as if it were solid
    d = np.zeros((3, 3), bool); d[0, 1] = d[1, 0] = d[1, 2] = d[2, 1] = True           # diamond of four diagonal links
    assert sg.outer_contour_area2(d) == 2 * 2


def cases():
    out = []
    # a thin frame (few pixels, large contour) beside a solid block (many pixels, smaller contour): pixel count picks the block
    m = np.zeros((60, 90), np.uint8); m[2:40, 2:50] = ring(38, 48, 2); m[42:58, 55:88] = 1
    out.append(("frame_vs_block", m))
    # the other way round: the solid block wins on both counts
    m = np.zeros((60, 90), np.uint8); m[2:20, 2:22] = ring(18, 20, 2); m[22:58, 30:88] = 1
    out.append(("block_wins", m))
    # an island inside the frame's hole, and a C shape (open frame: no hole, contour hugs the inside)
    m = np.zeros((70, 70), np.uint8); m[5:45, 5:45] = ring(40, 40, 3); m[15:30, 15:30] = 1; m[50:68, 10:60] = 1; m[53:65, 13:60] = 0
    out.append(("island_and_c", m))
    # two blobs of EQUAL contour area: the one discovered later in raster order wins (OpenCV's list is in reverse discovery order)
    m = np.zeros((60, 90), np.uint8); m[4:24, 6:36] = 1; m[30:50, 50:80] = 1
    out.append(("equal_blocks", m))
    m = np.zeros((60, 90), np.uint8); m[30:50, 6:36] = 1; m[4:24, 50:80] = 1; m[4:24, 40:42] = 1      # + a third, smaller one in between
    out.append(("equal_blocks_swapped", m))
    for s in range(4):
        out.append((f"random{s}", blobs(100 + s, 64, 80).astype(np.uint8)))
    out.append(("empty", np.zeros((20, 30), np.uint8)))
    out.append(("single_pixel", np.pad(np.ones((1, 1), np.uint8), 5)))
    return out


@pytest.mark.parametrize("name,mask", cases(), ids=[c[0] for c in cases()])
def test_library_host_tracer_equals_oracle(name, mask):
    assert _capi.host_largest_contour(mask) == sg.largest_component_box(mask)


def test_equal_areas_go_to_the_last_discovered():
    """np.argmax over cv2.findContours' list (reverse discovery order) keeps the LAST blob in raster order of first pixels."""
    by_name = dict(cases())
    assert sg.largest_component_box(by_name["equal_blocks"])[0] == (50, 30, 30, 20)
    assert sg.largest_component_box(by_name["equal_blocks_swapped"])[0] == (6, 30, 30, 20)
    assert _capi.host_largest_contour(by_name["equal_blocks"])[0] == (50, 30, 30, 20)


def test_frame_beats_block_unlike_pixel_count():
    name, m = cases()[0]
    (x, y, w, h), px = sg.largest_component_box(m)
    assert (x, y, w, h) == (2, 2, 48, 38) and px < 16 * 33                      # the frame: fewer pixels than the block, larger contour


@pytest.mark.gpu
@pytest.mark.parametrize("name,mask", cases(), ids=[c[0] for c in cases()])
def test_device_ranking_equals_oracle(name, mask):
    """sbbseg_page_box_dev = dilate x 6 + ranking; the oracle gets the same dilation (oracle/stage_glue.morph)."""
    import torch
    from gpu_common import make_model
    cfg, w, g, model = make_model(2, 64, 64, seed=0, precision="f16", max_batch=2)
    big = np.kron(mask, np.ones((3, 3), np.uint8))                                  # blow the shapes up so that the 25x25 dilation keeps holes open
    if name.startswith("random"):
        big = mask
    want = sg.page_box(big)
    d = torch.from_numpy(np.ascontiguousarray(big)).cuda()
    before = model.ctx.host_contour_calls()
    got = model.ctx.page_box_dev(d.data_ptr(), big.shape[0], big.shape[1])
    assert got == want, (name, got, want)
    model.ctx.set_conv_variant(1 << 21)                                             # force the host ranking: same answer
    assert model.ctx.page_box_dev(d.data_ptr(), big.shape[0], big.shape[1]) == want
    print(f"[{name}] host fallbacks: {model.ctx.host_contour_calls() - before} (1 = forced only)")
    model.release()
