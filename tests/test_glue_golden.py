"""The oracle's glue restatements and the product's glue, held to tests/golden/glue_golden.npz -- outputs of the REFERENCE's
own functions (``return_deskew_slope``, ``get_standard_deviation_of_summed_textline_patch_along_width``, ``otsu_copy``,
``get_image_and_scales``, ``extract_page``: main.py:1545-1718, 178-214, 384-437), produced by importing main.py with its cv2
calls bound to the oracle's OpenCV restatements (tests/golden/make_glue_golden.py).  Pinned: everything those functions do
around the cv2 calls; not pinned: the cv2 arithmetic itself ([EXT]).

CPU part: oracle == fixture, product host logic == fixture.  ``-m gpu`` part: device == fixture, through the C ABI."""
import os
import zlib

import numpy as np
import pytest

from oracle import deskew as dk
from oracle import stage_glue as sg
from oracle import tiling
from sbb_textline_detection_amd import predict, stages
from sbb_textline_detection_amd.synthetic import synthetic_page

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_golden.npz"))


def _patch(j):
    shp = tuple(int(v) for v in G[f"profile_patch_shape{j}"])
    return np.unpackbits(G[f"profile_patch{j}"])[:shp[0] * shp[1]].reshape(shp).astype(np.float64)


# ------------------------------------------------------------------------------------------- deskew (main.py:1545-1718)
@pytest.mark.parametrize("k", range(int(G["deskew_n"])))
def test_oracle_deskew_slope_equals_reference(k):
    assert dk.return_deskew_slope(G[f"deskew_mask{k}"], 1.0) == float(G[f"deskew_slope{k}"])


@pytest.mark.parametrize("k", range(int(G["deskew_n"])))
def test_product_host_sweep_equals_reference(k):
    """stages._deskew_sweep (the product's host half) over the ORACLE's row profiles: same angle as the reference."""
    m = G[f"deskew_mask{k}"]
    angles = np.linspace(-25, 25, 80)
    ang = stages._deskew_sweep(dk.row_profiles(m, angles), angles, 1.0)
    if abs(ang) > 15:
        angles = np.linspace(-90, -50, 30)
        ang = stages._deskew_sweep(dk.row_profiles(m, angles), angles, 1.0)
    assert ang == float(G[f"deskew_slope{k}"])


@pytest.mark.parametrize("j", range(int(G["profile_n"])))
def test_profile_statistics_equal_reference(j):
    _, _, sigma, mult = G[f"profile_case{j}"]
    y = _patch(j).sum(axis=1)
    for fn in (dk.profile_statistics, stages._profile_statistics):
        if int(G[f"profile_raises{j}"]):
            with pytest.raises(IndexError):
                fn(y, float(sigma), float(mult))
            continue
        lows, sd = fn(y, float(sigma), float(mult))
        assert np.array_equal(np.asarray(lows, np.float64), G[f"profile_lows{j}"]) and float(sd) == float(G[f"profile_std{j}"])


# ------------------------------------------------------------------------------------------- otsu_copy (main.py:178-194)
@pytest.mark.parametrize("k", range(int(G["otsu_n"])))
def test_otsu_copy_equals_reference(k):
    h, w, seed = (int(v) for v in G[f"otsu_case{k}"])
    page = synthetic_page(h, w, seed=seed)
    want = np.unpackbits(G[f"otsu_plane{k}"])[:h * w].reshape(h, w).astype(bool)
    for fn in (sg.otsu_copy, stages.otsu_copy):
        r = fn(page)
        assert r.dtype == np.float64 and r.shape == page.shape and set(np.unique(r)) <= {0.0, 255.0}
        assert np.array_equal(r[:, :, 0] > 0, want) and np.array_equal(r[:, :, 0], r[:, :, 1]) and np.array_equal(r[:, :, 0], r[:, :, 2])


# ------------------------------------------------------------------------------------------- get_image_and_scales (main.py:196-214)
@pytest.mark.parametrize("k", range(int(G["scale_n"])))
def test_scales_equal_reference(k):
    h, w, seed = (int(v) for v in G[f"scale_case{k}"])
    page = synthetic_page(max(h, 8), max(w, 8), seed=seed)[:h, :w]
    hi, wi, h_org, w_org = (int(v) for v in G[f"scale_result{k}"])
    assert stages.scaled_size(h, w) == (hi, wi) and (h_org, w_org) == (h, w)
    st = stages.InferenceStages("a", "b", "c")
    st.get_image_and_scales(page)
    assert (st.img_hight_int, st.img_width_int) == (hi, wi)
    assert [st.scale_y, st.scale_x] == [float(v) for v in G[f"scale_factors{k}"]]
    for fn in (tiling.resize_nearest, predict.resize_nearest):
        assert zlib.crc32(np.ascontiguousarray(fn(page, hi, wi)).tobytes()) & 0xFFFFFFFF == int(G[f"scale_crc{k}"])


# ------------------------------------------------------------------------------------------- extract_page glue (main.py:394-426)
def _border_mask(k):
    """The uint8 label image the fake border model leads do_prediction(patches=False) to (main.py:368-380)."""
    h, w, seed, mh, mw, bx, by, bw, bh, sy, sx = (int(v) for v in G[f"border_case{k}"])
    lab = np.zeros((mh, mw), np.uint8)
    lab[by:by + bh, bx:bx + bw] = 1
    lab[sy, sx] = 1
    return synthetic_page(h, w, seed=seed), tiling.resize_nearest(lab, h, w)


@pytest.mark.parametrize("k", range(int(G["border_n"])))
def test_page_box_and_crop_equal_reference(k):
    page, mask = _border_mask(k)
    box, pixels = sg.page_box(np.repeat(mask[:, :, None], 3, axis=2))
    crop, coord = sg.crop_image_inside_box(box, page)
    assert coord == [int(v) for v in G[f"border_coord{k}"]] and crop.shape == tuple(int(v) for v in G[f"border_crop_shape{k}"])
    assert zlib.crc32(np.ascontiguousarray(crop).tobytes()) & 0xFFFFFFFF == int(G[f"border_crop_crc{k}"])
    cont = np.array([[coord[2], coord[0]], [coord[3], coord[0]], [coord[3], coord[1]], [coord[2], coord[1]]])
    assert np.array_equal(cont, G[f"border_cont{k}"])
    hbox, hpix = stages.host_page_box(mask)                                   # the product's host mirror
    assert hbox == box and hpix == pixels


# ------------------------------------------------------------------------------------------- device == fixture
@pytest.fixture(scope="module")
def ctx():
    from gpu_common import make_model
    cfg, w, g, model = make_model(2, 64, 64, seed=0, precision="f16", max_batch=2)
    yield model.ctx
    model.release()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(int(G["deskew_n"])))
def test_device_deskew_slope_equals_reference(k, ctx):
    assert stages.return_deskew_slope(G[f"deskew_mask{k}"], 1.0, ctx=ctx) == float(G[f"deskew_slope{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(int(G["otsu_n"])))
def test_device_otsu_equals_reference(k):
    """sbbseg_segment_page_otsu's threshold + binarised gather: the threshold that reproduces the reference's plane."""
    from gpu_common import make_model
    h, w, seed = (int(v) for v in G[f"otsu_case{k}"])
    page = synthetic_page(h, w, seed=seed)
    cfg, wts, g, model = make_model(2, 224, 224, seed=1, precision="f16", max_batch=6)
    _, thr = model.ctx.segment_page_otsu(page)
    want = np.unpackbits(G[f"otsu_plane{k}"])[:h * w].reshape(h, w).astype(bool)
    assert np.array_equal(page[:, :, 0] > thr, want)
    model.release()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(int(G["border_n"])))
def test_device_page_box_equals_reference(k, ctx):
    import torch
    page, mask = _border_mask(k)
    d = torch.from_numpy(np.ascontiguousarray(mask)).cuda()
    box, pixels = ctx.page_box_dev(d.data_ptr(), mask.shape[0], mask.shape[1])
    y0, y1, x0, x1 = (int(v) for v in G[f"border_coord{k}"])
    assert tuple(box) == (x0, y0, x1 - x0, y1 - y0)


# ------------------------------------------------- get_text_region_contours_and_boxes (main.py:456-480; gate of main.py:2096)
def _regions(k):
    plane = G[f"regions_map{k}"]
    out = np.repeat(plane[:, :, None], 3, axis=2)
    if not int(G[f"regions_ch1_{k}"]):
        out[:, :, 1] = 0                                    # the case whose channels differ (np.all(image == (1, 1, 1)) fails)
    return out


@pytest.mark.parametrize("k", range(int(G["regions_n"])))
def test_oracle_text_region_contours_equal_reference(k):
    """The oracle's restatement keeps the contours the reference's own function kept (count and areas)."""
    assert sorted(sg.text_region_contour_areas(_regions(k))) == list(G[f"regions_areas{k}"])
    assert sg.text_regions_present(_regions(k)) == (len(G[f"regions_areas{k}"]) > 0)


@pytest.mark.parametrize("k", range(int(G["regions_n"])))
def test_product_host_text_region_gate_equals_reference(k):
    """stages.host_text_regions_present (foreign model objects; the library's host contour tracer) == `len(contours) > 0`."""
    assert stages.host_text_regions_present(_regions(k)) == (len(G[f"regions_areas{k}"]) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(int(G["regions_n"])))
def test_device_text_region_gate_equals_reference(k, ctx):
    """sbbseg_text_regions_present_dev (class mask, OPEN, CLOSE, contour ranking on the device) == `len(contours) > 0`."""
    r = _regions(k)
    if not int(G[f"regions_ch1_{k}"]):
        pytest.skip("the product's label planes have three equal channels by construction (one plane on the device)")
    assert ctx.text_regions_present(r) == (len(G[f"regions_areas{k}"]) > 0)
