"""``do_prediction`` -- the reference's patch-loop entry point (``main.py:225-380``), same signature
and return value, running on libsbbseg.

* uint8 page + :class:`~.model.SegModel`  ->  fused device path: one H2D copy of the page, LUT
  normalise + tiling + batched forward + argmax + stitch on the GPU, one D2H copy of the label map.
* anything else (float page, or a foreign model object with ``.layers``/``.predict``)  ->  the
  same tiling on the host with *batched* ``model.predict`` calls (the reference calls it with N=1).

The return value is ``uint8 [Hp, Wp, 3]`` with three identical channels, like the reference's.
"""
from __future__ import annotations

import numpy as np

from .model import SegModel


def _model_hwc(model):
    shp = model.layers[len(model.layers) - 1].output_shape          # main.py:227-229
    return int(shp[1]), int(shp[2]), int(shp[3])


def _axis(extent: int, tile: int, margin: int):
    """[(origin, crop_lo, crop_hi)] per tile of one axis (main.py:233-236, 246-281, 294-364)."""
    mid = tile - 2 * margin
    if extent < tile:
        raise ValueError(f"page extent {extent} smaller than model input {tile} "
                         "(the reference fails here too: main.py:278-281)")
    n = -(-extent // mid)
    out = []
    for t in range(n):
        d = min(t * mid, extent - tile)
        out.append((d, 0 if t == 0 else margin, tile if t == n - 1 else tile - margin))
    return out


def resize_nearest(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST) index rule (main.py:112-113); cv2 is not a
    dependency here.  src = min(floor(dst * (1/(dst_len/src_len))), src_len-1)  [OpenCV resizeNN]."""
    in_h, in_w = img.shape[:2]
    xs = np.minimum(np.floor(np.arange(out_w) * (1.0 / (out_w / float(in_w)))).astype(np.int64), in_w - 1)
    ys = np.minimum(np.floor(np.arange(out_h) * (1.0 / (out_h / float(in_h)))).astype(np.int64), in_h - 1)
    return img[ys][:, xs]


def _is_u8_image(img) -> bool:
    return isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3


def do_prediction(patches, img, model, full_image_shape=None, batch_size=None):
    """Drop-in for ``textline_detector.do_prediction(self, patches, img, model)``.

    ``full_image_shape`` plays the role of ``self.image.shape`` in the whole-image branch
    (main.py:378); default: ``img.shape``."""
    H, W, _ = _model_hwc(model)
    fused = isinstance(model, SegModel) and _is_u8_image(img)
    if patches:
        if fused:
            return model.segment_page(img, channels=3)                   # main.py:366 layout, built on the device
        x = np.asarray(img) / float(255.0)                               # main.py:239
        img_h, img_w = x.shape[0], x.shape[1]
        margin = int(0.1 * W)                                            # main.py:233 (width for both axes)
        xs, ys = _axis(img_w, W, margin), _axis(img_h, H, margin)
        tiles = [(x0, y0, xlo, xhi, ylo, yhi) for (x0, xlo, xhi) in xs for (y0, ylo, yhi) in ys]  # x outer
        out = np.zeros((img_h, img_w), np.uint8)
        bs = int(batch_size or getattr(model, "max_batch", 1))
        for k in range(0, len(tiles), bs):
            chunk = tiles[k:k + bs]
            batch = np.stack([x[y0:y0 + H, x0:x0 + W, :] for (x0, y0, *_r) in chunk])
            if isinstance(model, SegModel):
                probs = model.predict(batch)
            else:                                                        # foreign models keep N=1 (main.py:287)
                probs = np.concatenate([model.predict(b[None]) for b in batch])
            seg = np.argmax(probs, axis=3)                               # main.py:290
            for (x0, y0, xlo, xhi, ylo, yhi), s in zip(chunk, seg):
                out[y0 + ylo:y0 + yhi, x0 + xlo:x0 + xhi] = s[ylo:yhi, xlo:xhi]
        return np.repeat(out[:, :, np.newaxis], 3, axis=2)
    # patches == False: main.py:368-380
    shp = tuple(full_image_shape) if full_image_shape is not None else tuple(np.asarray(img).shape)
    if fused:
        return model.segment_whole(img, int(shp[0]), int(shp[1]), channels=3)
    x = resize_nearest(np.asarray(img) / float(255.0), H, W)
    probs = model.predict(x.reshape(1, x.shape[0], x.shape[1], x.shape[2]))
    seg = np.argmax(probs, axis=3)[0]
    seg3 = np.repeat(seg[:, :, np.newaxis], 3, axis=2)
    return resize_nearest(seg3, int(shp[0]), int(shp[1])).astype(np.uint8)


def do_prediction_pages(imgs, model):
    """``[do_prediction(True, img, model) for img in imgs]`` for a list of uint8 pages (main.py:490-503 over a batch of pages):
    same-sized pages go through the library's pipelined multi-page path (``sbbseg_segment_pages``: their tiles share
    full-sized chunks, and upload / compute / download overlap); anything else falls back to one call per page."""
    imgs = list(imgs)
    fused = isinstance(model, SegModel) and all(_is_u8_image(i) for i in imgs)
    if not fused or len({np.asarray(i).shape for i in imgs}) != 1 or len(imgs) < 2:
        return [do_prediction(True, i, model) for i in imgs]
    return model.ctx.segment_pages(imgs, channels=3)


class PatchSegmenter:
    """Carrier of the three reference methods that make up the hot path, with their original names
    and signatures, so reference-side code can be pointed here unchanged:

        start_new_session_and_model(model_dir) -> (model, session)      main.py:216-223
        do_prediction(patches, img, model) -> uint8 [H,W,3]              main.py:225-380
        resize_image(img, h, w)                                          main.py:112-113
    """

    def __init__(self, image=None, device: int = 0):
        self.image = image
        self.device = device

    def resize_image(self, img_in, input_height, input_width):
        return resize_nearest(img_in, input_height, input_width)

    def start_new_session_and_model(self, model_dir):
        from .model import start_new_session_and_model
        return start_new_session_and_model(model_dir, device=self.device)

    def do_prediction(self, patches, img, model):
        shp = self.image.shape if getattr(self, "image", None) is not None else None
        return do_prediction(patches, img, model, full_image_shape=shp)
