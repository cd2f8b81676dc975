"""The three stage wrappers that call the hot path and the steps either side of
it that section 8(f) ranks next, with the reference's names and call pattern:

    get_image_and_scales            main.py:196-214   upscale rule (2800 px high, or x1.2), INTER_NEAREST
    otsu_copy                       main.py:178-194   per-channel Otsu, channel-0 result in all 3 channels (quirk kept)
    extract_page   (model part)     main.py:384-392   border model, patches=False
    extract_text_regions            main.py:439-454   Otsu'd page, layout model (4 classes), patches=True
    textline_contours               main.py:490-503   textline model, patches=True, returns channel 0

    extract_page   (glue part)      main.py:394-426   border mask -> dilate x 6 -> largest blob -> box -> crop (device)
    erode x 3 / dilate x 4          main.py:2074-2075 on the layout map (device)

    return_deskew_slope             main.py:1601-1718 the per-region deskew search: rotate-and-project on the device (one launch
                                                      per sweep), the 1-D peak logic on the host with the reference's own scipy calls

The rest of the cv2 contour post-processing (text-region contours, line separation, ...) is out of scope.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .model import SegModel, start_new_session_and_model
from .predict import do_prediction


def scaled_size(h: int, w: int) -> Tuple[int, int]:
    """main.py:201-207: pages under 2500 px go to 2800 px height, everything else x1.2 (aspect kept)."""
    if h < 2500:
        hi = 2800
    else:
        hi = int(h * 1.2)
    return hi, int(hi * w / float(h))


def otsu_threshold_u8(ch: np.ndarray) -> int:
    """cv2.threshold(..., THRESH_BINARY + THRESH_OTSU) threshold value [EXT: OpenCV getThreshVal_Otsu_8u]:
    mean from the integer first moment times 1/N, probabilities h[i] * (1/N), strict `>` (first maximum).
    Host mirror of the device kernel (csrc/kernels.hip otsu_threshold_kernel), for foreign model objects."""
    h = np.bincount(np.ascontiguousarray(ch, np.uint8).reshape(-1), minlength=256)
    scale = 1.0 / float(h.sum())
    mu = 0.0
    for i in range(256):
        mu += float(i) * float(h[i])
    mu *= scale
    q1 = mu1 = best_sigma = 0.0
    best = 0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p_i = float(h[i]) * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + float(i) * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        d = mu1 - mu2
        sigma = q1 * q2 * d * d
        if sigma > best_sigma:
            best_sigma, best = sigma, i
    return best


def otsu_copy(img: np.ndarray) -> np.ndarray:
    """main.py:178-194.  Thresholds of all three channels are computed, but the channel-0 result is
    written to all three output channels (reference quirk, lines 191-193).  float64 HxWx3, values 0/255."""
    t = otsu_threshold_u8(np.ascontiguousarray(img[:, :, 0], np.uint8))
    b = np.where(img[:, :, 0] > t, 255.0, 0.0)
    return np.repeat(b[:, :, None], 3, axis=2)


def host_morph(plane: np.ndarray, is_max: bool, ksize: int, iterations: int) -> np.ndarray:
    """Host mirror of sbbseg_morph (foreign model objects only): one clipped (n(k-1)+1)-wide separable min / max."""
    r = (ksize - 1) // 2 * iterations
    a = np.ascontiguousarray(plane, np.uint8)
    fill = 0 if is_max else 255
    f = np.maximum if is_max else np.minimum
    for axis in (1, 0):
        p = np.pad(a, [(r, r) if ax == axis else (0, 0) for ax in (0, 1)], constant_values=fill)
        out = np.full_like(a, fill)
        for d in range(2 * r + 1):
            out = f(out, p[:, d:d + a.shape[1]] if axis == 1 else p[d:d + a.shape[0], :])
        a = out
    return a


def host_page_box(mask: np.ndarray):
    """Host mirror of sbbseg_page_box_dev (foreign model objects only): dilate x 6, then the component whose outer contour has
    the largest cv2.contourArea, traced by the library's host code (``sbbseg_debug_largest_contour``: no GPU involved)."""
    d = host_morph(np.where(np.asarray(mask) > 0, 255, 0).astype(np.uint8), True, 5, 6)
    from . import _capi
    return _capi.host_largest_contour(d)


def host_text_regions_present(regions: np.ndarray, label: int = 1, min_area: float = 0.00001) -> bool:
    """Host mirror of sbbseg_text_regions_present_dev (foreign model objects only; main.py:456-480): class mask -> OPEN -> CLOSE ->
    the largest outer-contour area (the library's host tracer, no GPU) >= min_area x H x W."""
    a = np.asarray(regions)
    m = np.all(a == label, axis=-1) if a.ndim == 3 else a == label
    p = np.where(m, 255, 0).astype(np.uint8)
    p = host_morph(host_morph(p, False, 5, 1), True, 5, 2)             # OPEN's erode + dilate, CLOSE's dilate ...
    p = host_morph(p, False, 5, 1)                                     # ... and erode
    if not p.any():
        return False
    from . import _capi
    return _capi.host_largest_contour_area2(p) * 0.5 >= min_area * float(p.shape[0] * p.shape[1])


def _profile_statistics(y: np.ndarray, sigma: float, multiplier: float):
    """get_standard_deviation_of_summed_textline_patch_along_width (main.py:1545-1599) from the row sums on: smoothed profile z,
    its maxima and the minima of the padded, negated profile (scipy, as the reference), the "deep" minima below
    mean(maxima > 10) * (1 - 1/multiplier), and std(z)."""
    from scipy.ndimage import gaussian_filter1d
    from scipy.signal import find_peaks
    y = np.asarray(y, np.float64)
    padded = np.zeros(len(y) + 20)
    padded[10:10 + len(y)] = y
    flipped = np.zeros(len(padded) + 20)
    flipped[10:10 + len(padded)] = padded.max() - padded
    z = gaussian_filter1d(y, sigma)
    minima = find_peaks(gaussian_filter1d(flipped, sigma), height=0)[0] - 20
    maxima = find_peaks(z, height=0)[0]
    tops = z[maxima]
    tops = tops[tops > 10]
    lows = z[minima]                                        # IndexError for a minimum in the right-hand padding: caller's except
    with np.errstate(all="ignore"):
        level = np.mean(tops) if tops.size else np.float64("nan")
    return lows[lows < level - level / multiplier], np.std(z)


def _deskew_sweep(profiles: np.ndarray, angles: np.ndarray, sigma: float) -> float:
    """One angle loop of return_deskew_slope (main.py:1630-1667): std of the smoothed profile per angle, arg max.  Kept: an angle
    without deep minima (their mean is NaN) is left out of the list, and the winner's position in the SHORTENED list indexes
    the full angle array (main.py:1655-1665)."""
    spread = []
    for k in range(len(angles)):
        try:
            lows, sd = _profile_statistics(profiles[k], sigma, 20.3)
            if lows.size == 0:
                continue                                    # np.mean([]) is NaN -> the reference skips the append
        except Exception:                                   # main.py:1652-1655
            sd = 0
        spread.append(sd)
    return float(angles[int(np.argmax(np.array(spread)))]) if spread else 0.0


def return_deskew_slope(img_patch: np.ndarray, sigma_des: float, ctx=None) -> float:
    """``textline_detector.return_deskew_slope`` (main.py:1601-1718): the rotation angle, out of 80 in [-25, 25] (and 30 in
    [-90, -50] when the first answer is steeper than 15 degrees), whose row profile of the rotated region mask varies most.
    The reference rotates the padded mask 80-110 times with cv2.warpAffine on the CPU (and forks cpu_count() processes over
    the regions, main.py:1760-1799); here one sweep is ONE launch (``sbbseg_deskew_profiles``).  ``ctx``: a
    ``_capi.Context`` (any finalized handle: the call does not touch the network)."""
    if ctx is None:
        raise RuntimeError("return_deskew_slope needs a library handle (SegModel.ctx): there is no CPU fallback")
    mask = np.ascontiguousarray(np.asarray(img_patch) != 0, np.uint8) if np.asarray(img_patch).dtype != np.uint8 else np.ascontiguousarray(img_patch)
    angles = np.linspace(-25, 25, 80)                                              # main.py:1622
    ang = _deskew_sweep(ctx.deskew_profiles(mask, angles), angles, sigma_des)
    if abs(ang) > 15:                                                              # main.py:1669-1670
        angles = np.linspace(-90, -50, 30)
        ang = _deskew_sweep(ctx.deskew_profiles(mask, angles), angles, sigma_des)
    return ang


class InferenceStages:
    """The model-running part of ``textline_detector.run()`` (main.py:2056-2107)."""

    def __init__(self, model_page_dir: str, model_region_dir: str, model_textline_dir: str, device: int = 0,
                 model_kwargs: Optional[dict] = None):
        self.model_page_dir, self.model_region_dir, self.model_textline_dir = model_page_dir, model_region_dir, model_textline_dir
        self.kw = dict(model_kwargs or {}, device=device)
        self.image = None
        self.scale_y = self.scale_x = 1.0

    def get_image_and_scales(self, image_u8: np.ndarray) -> None:
        """Records the upscaled geometry (main.py:196-214).  The upscaled page itself is not built:
        the tile gather reads the stored page through the nearest-neighbour index tables."""
        self.image_stored = np.ascontiguousarray(image_u8, np.uint8)
        h, w = image_u8.shape[:2]
        self.img_hight_int, self.img_width_int = scaled_size(h, w)
        self.scale_y = self.img_hight_int / float(h)
        self.scale_x = self.img_width_int / float(w)

    def _scaled_page(self):
        from .predict import resize_nearest
        return resize_nearest(self.image_stored, self.img_hight_int, self.img_width_int)

    def extract_page_mask(self) -> np.ndarray:
        """Border model on the whole page (main.py:384-392): uint8 [H,W,3] at the *scaled* size."""
        model, session = start_new_session_and_model(self.model_page_dir, **self.kw)
        try:
            if isinstance(model, SegModel):
                # stored page -> (virtual) upscaled page -> model size: the two nearest maps are composed on the
                # library side, the 4200x3000 page is never built on the host
                return model.ctx.segment_whole_scaled(self.image_stored, self.img_hight_int, self.img_width_int,
                                                      self.img_hight_int, self.img_width_int, channels=3)
            img = self._scaled_page()
            return do_prediction(False, img, model, full_image_shape=img.shape)
        finally:
            session.close()

    def extract_page(self):
        """main.py:384-437: border model + page box + crop.  Returns (croped_page, page_coord) like the reference;
        ``self.cont_page`` holds the box corners.  The mask never leaves the device between the model and the box.
        Like the reference, an empty border mask is an error (np.argmax of an empty list raises there, main.py:399-401)."""
        model, session = start_new_session_and_model(self.model_page_dir, **self.kw)
        try:
            if isinstance(model, SegModel):
                self.page_mask, box, pixels = model.ctx.extract_page_box(self.image_stored, self.img_hight_int, self.img_width_int, channels=3)
            else:
                from .predict import resize_nearest
                img = self._scaled_page()
                self.page_mask = do_prediction(False, img, model, full_image_shape=img.shape)
                box, pixels = host_page_box(self.page_mask[:, :, 0])
            if pixels == 0:
                raise ValueError("attempt to get argmax of an empty sequence")          # what main.py:401 raises
            x, y, w, h = box
            self.page_box = (x, y, w, h)
            # crop_image_inside_box (main.py:174-176) on the upscaled page; built here only for the crop
            page = self._scaled_page()
            croped_page, page_coord = page[y:y + h, x:x + w], [y, y + h, x, x + w]
            self.cont_page = [np.array([[page_coord[2], page_coord[0]], [page_coord[3], page_coord[0]],
                                        [page_coord[3], page_coord[1]], [page_coord[2], page_coord[1]]])]
            return croped_page, page_coord
        finally:
            session.close()

    def clean_text_regions(self, text_regions: np.ndarray) -> np.ndarray:
        """main.py:2074-2075: cv2.erode(text_regions, kernel, iterations=3) then cv2.dilate(..., iterations=4), on the device."""
        model, session = start_new_session_and_model(self.model_region_dir, **self.kw)
        try:
            plane = np.ascontiguousarray(text_regions[:, :, 0] if text_regions.ndim == 3 else text_regions, np.uint8)
            if isinstance(model, SegModel):
                out = model.ctx.morph(model.ctx.morph(plane, 0, 5, 3), 1, 5, 4)
            else:
                out = host_morph(host_morph(plane, False, 5, 3), True, 5, 4)
            return np.repeat(out[:, :, None], 3, axis=2) if text_regions.ndim == 3 else out
        finally:
            session.close()

    def extract_text_regions(self, img_u8: Optional[np.ndarray] = None, box=None) -> np.ndarray:
        """main.py:439-454.  On a SegModel the whole wrapper is one library call: histogram, Otsu threshold,
        binarising tile gather (and, with img_u8=None, the rescale of the stored page), forward, stitch.
        ``box`` = (x, y, w, h) of extract_page on the upscaled page: the stage runs on that crop (what run() hands it,
        main.py:2061-2072) without the crop being built."""
        model, session = start_new_session_and_model(self.model_region_dir, **self.kw)
        try:
            if isinstance(model, SegModel):
                if img_u8 is None and box is not None:
                    lab, self.otsu_threshold = model.ctx.segment_crop(self.image_stored, self.img_hight_int, self.img_width_int, box,
                                                                      binarise=True, channels=3)
                elif img_u8 is None:
                    lab, self.otsu_threshold = model.ctx.segment_page_otsu(self.image_stored, self.img_hight_int, self.img_width_int,
                                                                           channels=3)
                else:
                    lab, self.otsu_threshold = model.ctx.segment_page_otsu(np.ascontiguousarray(img_u8, np.uint8), channels=3)
                return lab                                                 # main.py:366 layout: 3 equal channels
            if img_u8 is None:
                img_u8 = self._scaled_page()
                if box is not None:
                    img_u8 = img_u8[box[1]:box[1] + box[3], box[0]:box[0] + box[2]]
            img = otsu_copy(img_u8).astype(np.uint8)                       # main.py:443-444
            return do_prediction(True, img, model)                         # main.py:447
        finally:
            session.close()            # (the reference's gc.collect() after every stage frees TF graphs; nothing to free here)

    def textline_contours(self, img_u8: Optional[np.ndarray] = None, box=None) -> np.ndarray:
        """main.py:490-503.  With img_u8=None the stored page is segmented through the fused rescale
        (identical to running on the upscaled page), restricted to ``box`` (extract_page's crop) when one is given."""
        model, session = start_new_session_and_model(self.model_textline_dir, **self.kw)
        try:
            if img_u8 is None and isinstance(model, SegModel):
                if box is not None:
                    return model.ctx.segment_crop(self.image_stored, self.img_hight_int, self.img_width_int, box)[0]
                return model.ctx.segment_page_scaled(self.image_stored, self.img_hight_int, self.img_width_int)
            if img_u8 is None:
                img_u8 = self._scaled_page()
                if box is not None:
                    img_u8 = img_u8[box[1]:box[1] + box[3], box[0]:box[0] + box[2]]
            return do_prediction(True, img_u8.astype(np.uint8), model)[:, :, 0]
        finally:
            session.close()            # (the reference's gc.collect() after every stage frees TF graphs; nothing to free here)

    def page_box_only(self):
        """extract_page (main.py:384-437) without building the cropped array: (page mask, (x, y, w, h), page_coord).  The mask
        stays on the device between the border model and the box search; an empty mask raises like main.py:401."""
        model, session = start_new_session_and_model(self.model_page_dir, **self.kw)
        try:
            if isinstance(model, SegModel):
                mask, box, pixels = model.ctx.extract_page_box(self.image_stored, self.img_hight_int, self.img_width_int, channels=3)
            else:
                img = self._scaled_page()
                mask = do_prediction(False, img, model, full_image_shape=img.shape)
                box, pixels = host_page_box(mask[:, :, 0])
            if pixels == 0:
                raise ValueError("attempt to get argmax of an empty sequence")
            x, y, w, h = box
            self.page_mask, self.page_box = mask, (x, y, w, h)
            page_coord = [y, y + h, x, x + w]
            self.cont_page = [np.array([[page_coord[2], page_coord[0]], [page_coord[3], page_coord[0]],
                                        [page_coord[3], page_coord[1]], [page_coord[2], page_coord[1]]])]
            return mask, self.page_box, page_coord
        finally:
            session.close()

    def text_regions_present(self, regions: np.ndarray) -> bool:
        """get_text_region_contours_and_boxes(...) returns at least one contour (main.py:456-480; run() gates the textline model on
        it, main.py:2083-2096): class-1 mask -> MORPH_OPEN -> MORPH_CLOSE -> a parentless contour of area >= 1e-5 x H x W.  On the
        device for a SegModel (``sbbseg_text_regions_present_dev``); foreign model objects get the host mirror."""
        model, session = start_new_session_and_model(self.model_region_dir, **self.kw)
        try:
            if isinstance(model, SegModel):
                return model.ctx.text_regions_present(regions, 1, 0.00001)
            return host_text_regions_present(regions)
        finally:
            session.close()

    def _run_resident(self):
        """run()'s three stages with the stored page uploaded ONCE and kept in device memory for all of them (run() hands the same
        page to every stage, main.py:2061-2102), the border mask and the region map staying on the device between their model and
        their glue: ONE library call, ``sbbseg_run_page`` (round 5: neither PyTorch nor device pointers on this path -- the
        reference's environment is Keras / TF only).  Same return value as the stage-by-stage path below, which remains the path for
        foreign model objects and for SBBSEG_STAGES_RESIDENT=0.  Returns None when it does not apply."""
        import os
        if os.environ.get("SBBSEG_STAGES_RESIDENT", "1") == "0":
            return None
        opened = [start_new_session_and_model(d, **self.kw) for d in (self.model_page_dir, self.model_region_dir, self.model_textline_dir)]
        try:
            (m_page, _), (m_region, _), (m_text, _) = opened
            if not all(isinstance(m, SegModel) for m in (m_page, m_region, m_text)):
                return None
            from . import _capi
            # extract_page's errors propagate (main.py:2061 is outside the try); the other stages degrade inside the call
            mask, regions, textlines, info = _capi.run_page(m_page.ctx, m_region.ctx, m_text.ctx, self.image_stored,
                                                            self.img_hight_int, self.img_width_int, channels=3)
            x, y, w, h = (int(v) for v in info.box_xywh)
            self.page_box = (x, y, w, h)
            page_coord = [y, y + h, x, x + w]
            self.cont_page = [np.array([[page_coord[2], page_coord[0]], [page_coord[3], page_coord[0]],
                                        [page_coord[3], page_coord[1]], [page_coord[2], page_coord[1]]])]
            self.page_mask = mask
            if info.regions_ok:
                self.otsu_threshold = int(info.otsu_threshold)
            return mask, regions, textlines, page_coord
        finally:
            for _, session in opened:
                session.close()

    def run(self, image_u8: np.ndarray):
        """The model-running part of run() (main.py:2056-2107) with its chaining: border model -> page box -> the layout
        and textline models on the CROPPED page (main.py:2061, 2072, 2102), text regions cleaned by erode x 3 / dilate x 4
        (main.py:2074-2075).  Returns (page mask [Hs,Ws,3], cleaned regions [h,w,3], text lines [h,w], page_coord) with
        h x w = extract_page's box; the crop itself is never built on a SegModel.  (Round 3 added page_coord as a FOURTH element:
        callers that unpacked three must be updated -- INTEGRATION.md.)  Like the reference, a failed layout stage yields
        regions = None and skips the textline model (textlines = None); extract_page's errors propagate."""
        self.get_image_and_scales(image_u8)
        resident = self._run_resident()
        if resident is not None:
            return resident
        page_mask, box, page_coord = self.page_box_only()          # outside the try, like main.py:2061: its errors propagate
        # main.py:2069-2091: the layout stage and its post-processing sit in a bare try/except -- any failure (the reference: a crop
        # smaller than the model input breaks do_prediction's reshape, main.py:278-285; here: sbbseg_segment_crop refuses it) degrades
        # to "no regions"; the textline model only runs when text regions were found (main.py:2096 `if len(contours) > 0`: contours
        # are traced from the opened / closed class-1 mask and filtered by area, main.py:456-480 -- tracing them is out of scope,
        # their existence is not: text_regions_present)
        try:
            regions = self.extract_text_regions(box=box)
            regions = self.clean_text_regions(regions)
            has_text = self.text_regions_present(regions)           # main.py:2083, 2096
        except Exception:
            regions, has_text = None, False
        textlines = None
        if has_text:
            try:
                textlines = self.textline_contours(box=box)
            except Exception:                                       # main.py:2152-2157: the outer bare except -> empty result
                textlines = None
        return page_mask, regions, textlines, page_coord
