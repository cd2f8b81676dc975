#!/usr/bin/env python3
"""Generate tests/golden/glue_golden.npz by running the REFERENCE's OWN glue functions (imported from /root/reference).

Runs only in the build container (needs /root/reference); never on the GPU box.  ``main.py`` is imported exactly as
``make_tiling_golden.py`` does it (empty stub modules for cv2 / keras / tensorflow / shapely / seaborn); scipy is real
(``gaussian_filter1d`` / ``find_peaks`` are the reference's own dependency).  The handful of cv2 entry points these
functions call are bound to the oracle's restatements of the OpenCV arithmetic:

    cv2.getRotationMatrix2D, cv2.warpAffine(INTER_CUBIC, BORDER_REPLICATE)  -> oracle/deskew.py
    cv2.threshold(THRESH_BINARY + THRESH_OTSU), cv2.threshold(x, 0, 255, 0) -> oracle/stage_glue.py
    cv2.resize(INTER_NEAREST), cv2.imread                                    -> oracle/tiling.py, a seeded page
    cv2.cvtColor(BGR2GRAY), cv2.dilate, cv2.findContours / contourArea / boundingRect -> oracle/stage_glue.py

What this pins: everything the reference does AROUND those calls -- padding geometry, the angle lists, the binarise-after-
rotate step, the padded / flipped profile, which minima count as "deep", the NaN / exception handling and the list-index
quirk of the sweep (main.py:1545-1718); the channel-0 quirk and dtypes of otsu_copy (main.py:178-194); the upscale rule and
scale factors (main.py:196-214); the border-mask -> box -> crop -> cont_page sequence of extract_page (main.py:394-426).
What it cannot pin: the OpenCV arithmetic inside the stubs ([EXT], cv2 is not installable) -- those stay "unpinned".

The fixture holds inputs (masks, pages) and the reference's outputs only.
    python tests/golden/make_glue_golden.py [out.npz]"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_tiling_golden import load_reference, stub_cv2_resize  # noqa: E402
from oracle import deskew as dk  # noqa: E402
from oracle import stage_glue as sg  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402  (seeded test pages: data, not product logic)


def text_mask(h, w, seed, slope, thick_div=3):
    """Seeded text-line-like region mask (0/1), lines following y = slope * x."""
    rng = np.random.RandomState(seed)
    m = np.zeros((h, w), np.uint8)
    period = max(8, h // 7)
    for y in range(period // 2, h - period // 2, period):
        x0, x1 = rng.randint(0, w // 6), w - rng.randint(0, w // 6)
        for x in range(x0, x1):
            yy = y + int(round(slope * (x - w / 2)))
            if 0 <= yy < h - period // thick_div:
                m[yy:yy + period // thick_div, x] = 1
    return m


DESKEW_MASKS = [(60, 90, 0, 0.05), (75, 48, 1, -0.12), (33, 33, 2, 0.0), (90, 140, 3, 0.30), (120, 80, 4, -0.45), (64, 64, 5, 0.0)]
# (mask index, rotation angle of the padded square, sigma, multiplier): the inputs return_deskew_slope hands over (main.py:1631-1640);
# the last one is an UNPADDED mask, where a minimum lands in the right-hand padding and the reference raises IndexError
PROFILE_CASES = [(0, 0.0, 1.0, 20.3), (1, -7.9, 1.0, 20.3), (3, 12.0, 2.0, 20.3), (5, 0.0, 0.5, 3.8), (4, -25.0, 1.5, 3.8), (0, None, 1.0, 3.8)]
OTSU_PAGES = [(300, 420, 0), (511, 333, 1), (256, 256, 2)]
SCALE_PAGES = [(180, 140, 3), (2600, 40, 4)]            # < 2500 high -> 2800; >= 2500 -> x 1.2
BORDER_CASES = [(300, 240, 7, (40, 30, 200, 150)), (280, 360, 8, (0, 0, 280, 360)), (200, 200, 9, (90, 60, 40, 100))]


def install_cv2_stubs(cv2, pages_by_name):
    cv2.INTER_NEAREST, cv2.INTER_CUBIC, cv2.BORDER_REPLICATE = 0, 2, 1
    cv2.THRESH_BINARY, cv2.THRESH_OTSU, cv2.COLOR_BGR2GRAY, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE = 0, 8, 6, 3, 2
    cv2.resize = stub_cv2_resize
    cv2.imread = lambda path: pages_by_name[path].copy()
    cv2.getRotationMatrix2D = lambda center, angle, scale: dk.rotation_matrix(center, angle)

    def warp_affine(src, M, dsize, flags=None, borderMode=None):
        assert flags == cv2.INTER_CUBIC and borderMode == cv2.BORDER_REPLICATE and tuple(dsize) == (src.shape[1], src.shape[0])
        return dk.warp_affine_cubic_replicate(src, M)
    cv2.warpAffine = warp_affine

    def threshold(src, thresh, maxval, typ):
        if typ == cv2.THRESH_BINARY + cv2.THRESH_OTSU:
            t = sg.otsu_threshold(np.ascontiguousarray(src, np.uint8))
            return float(t), np.where(src > t, maxval, 0).astype(np.uint8)
        assert typ == 0                                   # THRESH_BINARY with a fixed threshold (main.py:395)
        return float(thresh), np.where(src > thresh, maxval, 0).astype(np.uint8)
    cv2.threshold = threshold

    def cvt_color(img, code):
        assert code == cv2.COLOR_BGR2GRAY and np.array_equal(img[:, :, 0], img[:, :, 1]) and np.array_equal(img[:, :, 0], img[:, :, 2])
        return img[:, :, 0].copy()                        # equal channels: the fixed-point luma weights sum to one -> the value itself
    cv2.cvtColor = cvt_color
    cv2.dilate = lambda src, kernel, iterations=1: sg.morph(src, "dilate", kernel.shape[0], iterations)

    class Blob:                                           # stands in for one traced contour: its component's pixels
        def __init__(self, ys, xs):
            self.ys, self.xs = ys, xs

    def find_contours(mask, mode, method):
        from scipy import ndimage
        lab, n = ndimage.label(mask > 0, structure=np.ones((3, 3), int))
        # OpenCV's list order [EXT]: outer borders are discovered in raster order, each new contour is linked in at the head of its
        # parent's child list -> reverse discovery order (the reference's np.argmax then keeps the LAST-discovered of equal areas)
        return [Blob(*np.nonzero(lab == k)) for k in range(n, 0, -1)], None
    cv2.findContours = find_contours
    def contour_area(b):                                  # [EXT] the oracle's restatement: shoelace area of the traced outer border
        comp = np.zeros((int(b.ys.max()) + 1, int(b.xs.max()) + 1), bool)
        comp[b.ys, b.xs] = True
        return sg.outer_contour_area2(comp) / 2.0
    cv2.contourArea = contour_area
    cv2.boundingRect = lambda b: (int(b.xs.min()), int(b.ys.min()), int(b.xs.max() - b.xs.min() + 1), int(b.ys.max() - b.ys.min() + 1))


# (H, W, kind) -- region maps for get_text_region_contours_and_boxes; min_area 1e-5 x H x W is 16.8 px^2 on the 1400 x 1200 cases
TEXT_REGION_CASES = [(300, 240, 0), (300, 240, 1), (1400, 1200, 2), (1400, 1200, 3), (260, 300, 4), (200, 200, 5), (220, 260, 6), (1400, 1200, 7)]


def text_region_map(h, w, kind):
    """uint8 [h,w,3] layout label image (classes 0..3, three equal channels unless kind == 5)."""
    rng = np.random.RandomState(100 + kind)
    r = np.zeros((h, w), np.uint8)
    if kind == 0:                       # a text block, an image block of another class, specks that die in the opening
        r[40:160, 30:200] = 1
        r[180:260, 60:180] = 2
        for _ in range(12):
            y, x = rng.randint(0, h - 4), rng.randint(0, w - 4)
            r[y:y + rng.randint(1, 4), x:x + rng.randint(1, 4)] = 1
    elif kind == 1:                     # only specks (at most 4 x 4): nothing survives MORPH_OPEN
        for _ in range(30):
            y, x = rng.randint(0, h - 5), rng.randint(0, w - 5)
            r[y:y + rng.randint(1, 5), x:x + rng.randint(1, 5)] = 1
        r[100:104, 50:120] = 1          # a 4-pixel-high bar: thinner than the kernel
    elif kind == 2:                     # one 5 x 5 square: survives, contour area 16 < 16.8
        r[700:705, 600:605] = 1
    elif kind == 3:                     # one 5 x 6 rectangle: contour area 20 >= 16.8
        r[700:705, 600:606] = 1
    elif kind == 4:                     # ring with an island inside its hole (the island has a parent), and a separate block
        r[30:150, 30:170] = 1
        r[45:135, 45:155] = 0
        r[80:95, 90:110] = 1
        r[180:230, 200:280] = 1
    elif kind == 5:                     # class 1 in channel 0 only: np.all(image == (1, 1, 1)) is false everywhere
        r[50:150, 50:150] = 1
    elif kind == 6:                     # two blocks three pixels apart: MORPH_CLOSE joins them
        r[60:120, 40:100] = 1
        r[60:120, 103:170] = 1
        r[150:200, 40:170] = 3
    elif kind == 7:                     # thin L (5 wide) and a 4-wide bar next to it; other classes around
        r[300:305, 200:260] = 1
        r[300:340, 200:205] = 1
        r[500:504, 200:400] = 1
        r[800:1000, 300:900] = 2
    out = np.repeat(r[:, :, None], 3, axis=2)
    if kind == 5:
        out[:, :, 1] = 0
    return out


def _ring_area(c):
    pts = np.asarray(c, np.float64).reshape(-1, 2)
    x, y = pts[:, 0], pts[:, 1]
    return abs(float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))) / 2.0


def install_contour_tree_stubs(cv2, ref):
    """cv2.morphologyEx / findContours(RETR_TREE) / boundingRect on point arrays, shapely's Polygon -- for main.py:456-480."""
    from scipy import ndimage
    cv2.MORPH_OPEN, cv2.MORPH_CLOSE = 2, 3
    cv2.cv2 = cv2                                          # main.py:471 writes cv2.cv2.RETR_TREE

    def morphology_ex(src, op, kernel):
        k = kernel.shape[0]
        planes = [src[:, :, c] for c in range(src.shape[2])] if src.ndim == 3 else [src]
        res = []
        for pl in planes:
            if op == cv2.MORPH_OPEN:
                res.append(sg.morph(sg.morph(pl, "erode", k, 1), "dilate", k, 1))
            else:
                assert op == cv2.MORPH_CLOSE
                res.append(sg.morph(sg.morph(pl, "dilate", k, 1), "erode", k, 1))
        return np.stack(res, axis=2) if src.ndim == 3 else res[0]
    cv2.morphologyEx = morphology_ex

    def approx_simple(chain):
        n = len(chain)
        if n < 3:
            return chain
        keep = []
        for i in range(n):
            (x0, y0), (x1, y1), (x2, y2) = chain[i - 1], chain[i], chain[(i + 1) % n]
            if (x1 - x0, y1 - y0) != (x2 - x1, y2 - y1):
                keep.append(chain[i])
        return keep

    def find_contours(mask, mode, method):
        assert mode == cv2.RETR_TREE and method == cv2.CHAIN_APPROX_SIMPLE
        fg = mask > 0
        lab, n = ndimage.label(fg, structure=np.ones((3, 3), int))
        # [EXT, a restatement -- NOT OpenCV] RETR_TREE as a list: the outer border of every 8-connected component as the oracle traces it,
        # followed by the borders of its holes (the component's pixels around each hole); hierarchy[..][3] = parent: -1 for a component
        # that touches the outer background, the hole it lies in for an island, the component for a hole.  Components in reverse discovery
        # order, as in the extract_page stub.  Hole contours are in the list since round 6: they never pass the reference's
        # `parent == -1` test, but they do advance its `jv` counter (main.py:80-92), which indexes the hierarchy by the contours that were
        # NOT skipped for having fewer than three points -- with them in the list a skipped contour would shift every later lookup.
        bg = np.pad(~fg, 1, constant_values=True)
        blab, _nb = ndimage.label(bg, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        outer_bg = blab == blab[0, 0]
        pl = np.pad(lab, 1, constant_values=0)
        touch = np.zeros(n + 1, bool)
        for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
            sh = np.roll(outer_bg, (dy, dx), axis=(0, 1))
            touch[np.unique(pl[sh & (pl > 0)])] = True
        hole_lab = blab[1:-1, 1:-1].copy()
        hole_lab[outer_bg[1:-1, 1:-1]] = 0                 # > 0: a 4-connected background region enclosed by foreground
        order = list(range(n, 0, -1))

        def trace(region):
            ys, xs = np.nonzero(region)
            y0, x0 = ys.min(), xs.min()
            chain = sg.outer_contour_chain(region[y0:ys.max() + 1, x0:xs.max() + 1])
            return np.array([[[x + x0, y + y0]] for (x, y) in approx_simple(chain)], np.int32)
        # holes of each component: the enclosed background regions lying inside its filled shape and bordered by its pixels
        filled = {k: ndimage.binary_fill_holes(lab == k) for k in order}
        holes_of = {k: [] for k in order}
        for hid in [int(h) for h in np.unique(hole_lab) if h > 0]:
            hole = hole_lab == hid
            ring = ndimage.binary_dilation(hole, structure=np.ones((3, 3), bool)) & fg
            owners = [int(o) for o in np.unique(lab[ring]) if filled[int(o)][hole].all()]
            assert len(owners) == 1
            holes_of[owners[0]].append((hid, ring & (lab == owners[0])))
        contours, hier, index_of_comp, index_of_hole = [], [], {}, {}
        for k in order:
            index_of_comp[k] = len(contours)
            contours.append(trace(lab == k))
            hier.append([-1, -1, -1, -1])
            for hid, ring in holes_of[k]:
                index_of_hole[hid] = len(contours)
                contours.append(trace(ring))
                hier.append([-1, -1, -1, index_of_comp[k]])
        for k in order:                                    # an island's parent is the hole it lies in
            if not touch[k]:
                around = ndimage.binary_dilation(lab == k, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]]) & ~fg
                hid = int(hole_lab[around].max())
                assert hid > 0
                hier[index_of_comp[k]][3] = index_of_hole[hid]
        find_contours.last = (len(contours), sum(1 for h in hier if h[3] >= 0), min((len(c) for c in contours), default=0))
        return contours, np.array([hier], np.int32)
    cv2.findContours = find_contours

    old_rect = cv2.boundingRect

    def bounding_rect(c):
        if isinstance(c, np.ndarray):
            p = np.asarray(c, np.int64).reshape(-1, 2)
            return (int(p[:, 0].min()), int(p[:, 1].min()), int(p[:, 0].max() - p[:, 0].min() + 1), int(p[:, 1].max() - p[:, 1].min() + 1))
        return old_rect(c)
    cv2.boundingRect = bounding_rect

    class Polygon:                                        # shapely.geometry.Polygon of a ring: |shoelace| / 2, closed exterior
        def __init__(self, pts):
            self.pts = [tuple(float(v) for v in p) for p in pts]
            self.area = _ring_area(self.pts)
            self.exterior = type("E", (), {"coords": self.pts + self.pts[:1]})()
    ref.geometry.Polygon = Polygon


class BorderModel:
    """Fake border model: class 1 inside a seeded rectangle of the MODEL-sized input (plus a one-pixel speck elsewhere)."""

    class _L:
        def __init__(self, shp):
            self.output_shape = shp

    def __init__(self, H, W, box, speck):
        self.layers = [self._L((None, H, W, 2))]
        self.H, self.W, self.box, self.speck = H, W, box, speck

    def predict(self, x):
        p = np.zeros((1, self.H, self.W, 2), np.float32)
        p[..., 0] = 1.0
        y, x0, h, w = self.box[1], self.box[0], self.box[3], self.box[2]
        p[0, y:y + h, x0:x0 + w, 1] = 2.0
        p[0, self.speck[0], self.speck[1], 1] = 2.0
        return p


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "glue_golden.npz")
    ref = load_reference()
    import cv2
    pages = {}
    install_cv2_stubs(cv2, pages)
    det = ref.textline_detector.__new__(ref.textline_detector)
    det.kernel = np.ones((5, 5), np.uint8)                # main.py:57
    out = {}

    # ---- return_deskew_slope / get_standard_deviation_of_summed_textline_patch_along_width (main.py:1545-1718)
    masks = [text_mask(*c) for c in DESKEW_MASKS]
    for k, m in enumerate(masks):
        out[f"deskew_mask{k}"] = m
        out[f"deskew_slope{k}"] = np.float64(det.return_deskew_slope(m, 1.0))
        print("deskew", k, m.shape, float(out[f"deskew_slope{k}"]))
    out["deskew_n"] = np.int64(len(masks))
    for j, (mi, angle, sigma, mult) in enumerate(PROFILE_CASES):
        out[f"profile_case{j}"] = np.array([mi, np.nan if angle is None else angle, sigma, mult], np.float64)
        if angle is None:
            patch = masks[mi].astype(np.float64)
        else:
            patch = det.rotate_image(dk.padded_square(masks[mi]), angle)          # the reference's own method over the stubs
            patch[patch != 0] = 1                                                  # main.py:1633
        out[f"profile_patch{j}"] = np.packbits(patch != 0)
        out[f"profile_patch_shape{j}"] = np.array(patch.shape, np.int64)
        try:
            lows, sd = det.get_standard_deviation_of_summed_textline_patch_along_width(patch, sigma, mult)
            out[f"profile_raises{j}"] = np.int64(0)
        except IndexError:                                # a minimum found in the right-hand padding (main.py:1586): the sweep's
            lows, sd = [], 0.0                            # except clause turns it into var_spectrum = 0 (main.py:1652-1655)
            out[f"profile_raises{j}"] = np.int64(1)
        out[f"profile_lows{j}"] = np.asarray(lows, np.float64)
        out[f"profile_std{j}"] = np.float64(sd)
        print("profile", j, len(lows), float(sd), "raises" if out[f"profile_raises{j}"] else "")
    out["profile_n"] = np.int64(len(PROFILE_CASES))

    # ---- otsu_copy (main.py:178-194): the page itself is regenerated from its seed by the tests
    for k, (h, w, seed) in enumerate(OTSU_PAGES):
        page = synthetic_page(h, w, seed=seed)
        r = det.otsu_copy(page)
        assert r.dtype == np.float64 and r.shape == page.shape
        assert np.array_equal(r[:, :, 0], r[:, :, 1]) and np.array_equal(r[:, :, 0], r[:, :, 2])
        out[f"otsu_case{k}"] = np.array([h, w, seed], np.int64)
        out[f"otsu_plane{k}"] = np.packbits(r[:, :, 0] > 0)
        out[f"otsu_values{k}"] = np.unique(r)
        print("otsu", k, page.shape, np.unique(r), float((r[:, :, 0] > 0).mean()))
    out["otsu_n"] = np.int64(len(OTSU_PAGES))

    # ---- get_image_and_scales (main.py:196-214)
    for k, (h, w, seed) in enumerate(SCALE_PAGES):
        page = synthetic_page(max(h, 8), max(w, 8), seed=seed)[:h, :w]
        pages[f"page{k}"] = page
        det.image_dir = f"page{k}"
        det.get_image_and_scales()
        out[f"scale_case{k}"] = np.array([h, w, seed], np.int64)
        out[f"scale_result{k}"] = np.array([det.img_hight_int, det.img_width_int, det.height_org, det.width_org], np.int64)
        out[f"scale_factors{k}"] = np.array([det.scale_y, det.scale_x], np.float64)
        out[f"scale_crc{k}"] = np.int64(zlib.crc32(np.ascontiguousarray(det.image).tobytes()) & 0xFFFFFFFF)
        print("scale", k, (h, w), "->", det.image.shape, det.scale_y, det.scale_x)
    out["scale_n"] = np.int64(len(SCALE_PAGES))

    # ---- extract_page glue (main.py:384-437) around a fake border model
    for k, (h, w, seed, box) in enumerate(BORDER_CASES):
        page = synthetic_page(h, w, seed=seed)
        det.image = page
        mh = mw = 64
        bx = (box[0] * mw // w, box[1] * mh // h, max(1, box[2] * mw // w), max(1, box[3] * mh // h))
        fm = BorderModel(mh, mw, bx, (mh - 2, 1))
        det.start_new_session_and_model = lambda _dir, fm=fm: (fm, type("S", (), {"close": lambda self: None})())
        det.model_page_dir = "unused"
        croped, coord = det.extract_page()
        out[f"border_case{k}"] = np.array([h, w, seed, mh, mw, *bx, mh - 2, 1], np.int64)
        out[f"border_coord{k}"] = np.array(coord, np.int64)
        out[f"border_cont{k}"] = np.asarray(det.cont_page[0], np.int64)
        out[f"border_crop_shape{k}"] = np.array(croped.shape, np.int64)
        out[f"border_crop_crc{k}"] = np.int64(zlib.crc32(np.ascontiguousarray(croped).tobytes()) & 0xFFFFFFFF)
        print("border", k, (h, w), "box", bx, "->", coord, croped.shape)
    out["border_n"] = np.int64(len(BORDER_CASES))
    # ---- get_text_region_contours_and_boxes (main.py:456-480): the `len(contours) > 0` that gates the textline model (main.py:2096)
    install_contour_tree_stubs(cv2, ref)
    for k, (h, w, kind) in enumerate(TEXT_REGION_CASES):
        regions = text_region_map(h, w, kind)
        kept = det.get_text_region_contours_and_boxes(regions)
        areas = sorted(_ring_area(c) for c in kept)
        assert len(det.boxes) == len(kept)
        out[f"regions_case{k}"] = np.array([h, w, kind], np.int64)
        out[f"regions_map{k}"] = regions[:, :, 0].copy()
        out[f"regions_ch1_{k}"] = np.int64(int(np.array_equal(regions[:, :, 0], regions[:, :, 1])))
        out[f"regions_areas{k}"] = np.asarray(areas, np.float64)
        print("regions", k, (h, w), "kind", kind, "->", len(kept), "contours", areas[:4])
    out["regions_n"] = np.int64(len(TEXT_REGION_CASES))
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
