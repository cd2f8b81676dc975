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
