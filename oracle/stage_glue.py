"""ORACLE (test infrastructure, not product code) -- CPU restatement of the steps either side of the
hot path in the reference's stage wrappers:

    otsu_copy                 /root/reference/qurator/sbb_textline_detector/main.py:178-194
    extract_text_regions      main.py:439-447   (otsu_copy -> astype(uint8) -> do_prediction(patches=True))

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

PARITY UNPINNED: the arithmetic lives in OpenCV (``cv2.threshold(..., THRESH_BINARY + THRESH_OTSU)``;
opencv-python-headless, unpinned in the reference's requirements.txt), which is not installed in this
image and has no test vectors in the reference.  ``otsu_threshold`` restates the published algorithm of
OpenCV's ``getThreshVal_Otsu_8u`` (modules/imgproc/src/thresh.cpp, 3.x/4.x) [EXT]: same loop, same fp64
operation order (mean from the integer moment times 1/N, probabilities as h[i] * (1/N), strict `>`
so the FIRST maximum of the between-class variance wins).  What IS pinned is the reference's own
code around it: channel 0's binarisation is written to all three output channels (main.py:191-193).
"""
from __future__ import annotations

import numpy as np

FLT_EPSILON = float(np.finfo(np.float32).eps)


def histogram_u8(ch: np.ndarray) -> np.ndarray:
    return np.bincount(np.ascontiguousarray(ch, np.uint8).reshape(-1), minlength=256).astype(np.int64)


def otsu_threshold_from_hist(hist: np.ndarray) -> int:
    """getThreshVal_Otsu_8u [EXT] on a 256-bin histogram; plain Python floats = IEEE fp64, no FMA."""
    h = [int(v) for v in hist]
    n = sum(h)
    scale = 1.0 / float(n)
    mu = 0.0
    for i in range(256):
        mu += float(i) * float(h[i])
    mu *= scale
    mu1 = 0.0
    q1 = 0.0
    max_sigma = 0.0
    max_val = 0
    for i in range(256):
        p_i = float(h[i]) * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < FLT_EPSILON or max(q1, q2) > 1.0 - FLT_EPSILON:
            continue
        mu1 = (mu1 + float(i) * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        d = mu1 - mu2
        sigma = q1 * q2 * d * d
        if sigma > max_sigma:
            max_sigma = sigma
            max_val = i
    return max_val


def otsu_threshold(ch: np.ndarray) -> int:
    return otsu_threshold_from_hist(histogram_u8(ch))


def otsu_copy(img: np.ndarray) -> np.ndarray:
    """main.py:178-194: float64 [H,W,3]; every channel holds the binarisation of channel 0
    (THRESH_BINARY: src > thresh ? 255 : 0)."""
    t = otsu_threshold(img[:, :, 0])
    b = np.where(np.asarray(img[:, :, 0], np.int64) > t, 255.0, 0.0)
    out = np.zeros(img.shape)                       # main.py:179
    out[:, :, 0] = b                                # main.py:191-193 (threshold1 three times)
    out[:, :, 1] = b
    out[:, :, 2] = b
    return out


# ------------------------------------------------------------------------------------------------------------------
# Morphology and page box (SURVEY.md 8f-3 remainder).  Reference: self.kernel = np.ones((5, 5), np.uint8) (main.py:57);
#   extract_page           main.py:394-404   cvtColor -> threshold(>0 -> 255) -> dilate x 6 -> findContours -> largest -> boundingRect
#   textline_contours      main.py:2074-2075 text_regions = erode x 3, then dilate x 4
#   crop_image_inside_box  main.py:174-176
# cv2.erode / cv2.dilate semantics restated [EXT OpenCV morph.dispatch.cpp]: flat structuring element, anchor at its centre,
# borderType = BORDER_CONSTANT with borderValue = morphologyDefaultBorderValue() -- +inf for erode, -inf for dilate, i.e. pixels
# outside the image never win the min / max; `iterations` applies the filter repeatedly.  Written here LITERALLY (one 5x5 pass
# per iteration), so the device's shortcut (one clipped (4n+1)-wide separable filter) is checked, not assumed.
# cv2.findContours / contourArea / boundingRect: restated as border following + shoelace area below (largest_component_box);
# see include/sbbseg.h sbbseg_page_box_dev for what that leaves [EXT, unpinned].
def _morph_once(a: np.ndarray, k: int, is_max: bool) -> np.ndarray:
    r = (k - 1) // 2
    fill = 0 if is_max else 255
    p = np.pad(a, r, constant_values=fill)
    out = np.full_like(a, fill)
    H, W = a.shape
    for dy in range(k):
        for dx in range(k):
            v = p[dy:dy + H, dx:dx + W]
            out = np.maximum(out, v) if is_max else np.minimum(out, v)
    return out


def morph(plane: np.ndarray, op: str, ksize: int = 5, iterations: int = 1) -> np.ndarray:
    """cv2.erode (op='erode') / cv2.dilate (op='dilate') of a uint8 plane with a ksize x ksize kernel of ones."""
    a = np.ascontiguousarray(plane, np.uint8)
    for _ in range(iterations):
        a = _morph_once(a, ksize, op == "dilate")
    return a


def region_cleanup(text_regions: np.ndarray) -> np.ndarray:
    """main.py:2074-2075 on the layout stage's label image (any number of equal channels)."""
    if text_regions.ndim == 3:
        return np.stack([region_cleanup(text_regions[:, :, c]) for c in range(text_regions.shape[2])], axis=2)
    return morph(morph(text_regions, "erode", 5, 3), "dilate", 5, 4)


_DX8 = (1, 1, 0, -1, -1, -1, 0, 1)          # E, SE, S, SW, W, NW, N, NE: clockwise with y pointing down
_DY8 = (0, 1, 1, 1, 0, -1, -1, -1)


def outer_contour_chain(comp: np.ndarray):
    """The OUTER border cv2.findContours traces around the 8-connected component ``comp`` (bool [H,W], exactly one component) as
    the closed chain of boundary pixels [(x, y), ...] (CHAIN_APPROX_NONE; CHAIN_APPROX_SIMPLE drops the collinear ones) [EXT:
    OpenCV's border following].  Moore neighbour tracing from the first pixel in raster order (its west neighbour is background),
    stopped when the start pixel is left again in the first direction.  A single pixel gives a one-point chain."""
    H, W = comp.shape
    ys, xs = np.nonzero(comp)
    sy, sx = int(ys[0]), int(xs[0])                      # np.nonzero is row-major: the first pixel in raster order

    def inside(y, x):
        return 0 <= y < H and 0 <= x < W and bool(comp[y, x])
    cy, cx, back, first = sy, sx, 4, None
    chain = [(sx, sy)]
    for _ in range(4 * H * W + 8):
        d = None
        for k in range(1, 9):
            dd = (back + k) & 7
            if inside(cy + _DY8[dd], cx + _DX8[dd]):
                d = dd
                break
        if d is None:
            return chain                                 # a single pixel
        if (cy, cx) == (sy, sx):
            if first is None:
                first = d
            elif d == first:
                break
        cy, cx = cy + _DY8[d], cx + _DX8[d]
        chain.append((cx, cy))
        back = (d + (5 if d & 1 else 6)) & 7             # the background neighbour examined just before, seen from the new pixel
    return chain[:-1]                                    # the last step re-entered the start pixel


def outer_contour_area2(comp: np.ndarray) -> int:
    """TWICE the area cv2.contourArea gives for that outer contour: the shoelace sum over the closed chain of pixel centres."""
    c = outer_contour_chain(comp)
    area2 = 0
    for k in range(len(c)):
        (x0, y0), (x1, y1) = c[k], c[(k + 1) % len(c)]
        area2 += x0 * y1 - x1 * y0
    return abs(area2)


def largest_component_box(mask: np.ndarray):
    """((x, y, w, h), pixels) of the 8-connected component of mask > 0 whose OUTER CONTOUR has the largest cv2.contourArea
    (main.py:398-404: contours[np.argmax([cv2.contourArea(c) ...])], cv2.boundingRect) -- a hole's contour never wins, it lies
    inside its component's outer contour; ((0,0,0,0), 0) if the mask is empty.  Ties: the LAST component in raster order of first
    pixels [EXT, restated from OpenCV's contours.cpp, unpinned: outer borders are discovered in raster order and every new contour
    is linked in at the HEAD of its parent's child list (cvInsertNodeIntoTree), so cv2.findContours returns siblings in reverse
    discovery order and np.argmax's first maximum (main.py:400-401) is the last-discovered of equal areas]."""
    from scipy import ndimage
    lab, n = ndimage.label(np.asarray(mask) > 0, structure=np.ones((3, 3), int))
    if n == 0:
        return (0, 0, 0, 0), 0
    slices = ndimage.find_objects(lab)
    areas = [outer_contour_area2(lab[sl] == k + 1) for k, sl in enumerate(slices)]
    # scipy numbers components in raster order of their first pixel = OpenCV's discovery order; the reference's list is reversed
    best = len(areas) - 1 - int(np.argmax(areas[::-1]))  # ties -> the last discovered
    sl = slices[best]
    pixels = int((lab[sl] == best + 1).sum())
    return (int(sl[1].start), int(sl[0].start), int(sl[1].stop - sl[1].start), int(sl[0].stop - sl[0].start)), pixels


def page_box(img_page_prediction: np.ndarray):
    """main.py:394-404 from do_prediction(patches=False)'s uint8 [H,W,3] (or [H,W]) label image.  BGR2GRAY of three equal
    channels returns the channel (the weights sum to one); threshold(gray, 0, 255, THRESH_BINARY) = gray > 0."""
    gray = img_page_prediction[:, :, 0] if img_page_prediction.ndim == 3 else img_page_prediction
    thresh = np.where(gray > 0, 255, 0).astype(np.uint8)             # main.py:394-395
    thresh = morph(thresh, "dilate", 5, 6)                           # main.py:397
    return largest_component_box(thresh)                             # main.py:398-404 (see [EXT] notes there)


def crop_image_inside_box(box, img):
    """main.py:174-176."""
    x, y, w, h = box
    return img[y:y + h, x:x + w], [y, y + h, x, x + w]


# ------------------------------------------------------------------------------------------------------------------
# get_text_region_contours_and_boxes (main.py:456-480): run() calls the textline model only when this returns at least
# one contour (main.py:2083, 2096 `if len(contours) > 0`).  Restated down to the list of kept contours' areas; the
# polygons themselves (CHAIN_APPROX_SIMPLE point lists, boundingRect) are outside the hot path's scope.
#   mask_texts = np.all(image == (1, 1, 1), axis=-1)  -> * 255 -> uint8                      main.py:457-461
#   cv2.morphologyEx(MORPH_OPEN, kernel) = dilate(erode(x)); MORPH_CLOSE = erode(dilate(x))    main.py:463-464  [EXT]
#   cvtColor(BGR2GRAY) of three equal channels = the channel; threshold(imgray, 0, 255, 0) = > 0     main.py:467-469
#   cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) [EXT Suzuki-Abe border following]: one outer border per
#     8-connected component, one hole border per 4-connected background component that does not reach the frame;
#     hierarchy[0][j][3] == -1 <=> contour j is the outer border of a component that lies in the frame-connected background
#   filter_contours_area_of_image(thresh, contours, hierarchy, max_area=1, min_area=0.00001)    main.py:77-92:
#     keep contour j when len(c) >= 3, min_area * H * W <= shapely Polygon(c).area <= max_area * H * W and it has no parent.
#     (Polygon(...).area of a ring = |shoelace sum| / 2 = cv2.contourArea; dropping collinear points does not change it.)
# Quirk of the reference NOT restated: contours with fewer than three points `continue` past `jv += 1` (main.py:82-83),
# after which hierarchy[0][jv] belongs to an earlier contour.  It cannot fire here: after OPEN and CLOSE every component
# and every hole is a union of 5x5 squares, whose border has at least four corner points.
def text_region_contour_areas(regions: np.ndarray, label: int = 1, min_area: float = 0.00001, max_area: float = 1.0):
    """Areas (cv2.contourArea units) of the contours get_text_region_contours_and_boxes keeps, in raster order of the
    components' first pixels.  ``regions``: uint8 [H,W,3] (all three channels are compared, main.py:457-458) or [H,W]."""
    from scipy import ndimage
    a = np.asarray(regions)
    m = np.all(a == label, axis=-1) if a.ndim == 3 else (a == label)
    img = np.where(m, 255, 0).astype(np.uint8)
    img = morph(morph(img, "erode", 5, 1), "dilate", 5, 1)               # MORPH_OPEN
    img = morph(morph(img, "dilate", 5, 1), "erode", 5, 1)               # MORPH_CLOSE
    fg = img > 0
    H, W = fg.shape
    lab, n = ndimage.label(fg, structure=np.ones((3, 3), int))           # 8-connected foreground
    if n == 0:
        return []
    # frame-connected background: 4-connected flood of the complement from a one-pixel frame around the image
    bg = np.pad(~fg, 1, constant_values=True)
    blab, _ = ndimage.label(bg, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outer_bg = (blab == blab[0, 0])
    touch = np.zeros(n + 1, bool)                                        # component k is 4-adjacent to the frame-connected background
    pl = np.pad(lab, 1, constant_values=0)
    for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
        sh = np.roll(outer_bg, (dy, dx), axis=(0, 1))
        touch[np.unique(pl[sh & (pl > 0)])] = True
    kept = []
    total = float(np.prod(fg.shape[:2]))
    for k, sl in enumerate(ndimage.find_objects(lab)):
        area = outer_contour_area2(lab[sl] == k + 1) / 2.0
        if min_area * total <= area <= max_area * total and touch[k + 1]:
            kept.append(area)
    return kept


def text_regions_present(regions: np.ndarray, label: int = 1, min_area: float = 0.00001) -> bool:
    """`len(contours) > 0` of main.py:2096 for the cleaned layout map."""
    return len(text_region_contour_areas(regions, label, min_area)) > 0
