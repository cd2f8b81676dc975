"""CPU restatement of the deskew search -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke, bench cpu leg).

Follows /root/reference/qurator/sbb_textline_detector/main.py:
  * 159-163    rotate_image: cv2.getRotationMatrix2D((w // 2, h // 2), slope, 1.0) + cv2.warpAffine(INTER_CUBIC, BORDER_REPLICATE)
  * 1545-1599  get_standard_deviation_of_summed_textline_patch_along_width: row sums, Gaussian smoothing, peak logic
  * 1601-1718  return_deskew_slope: pad to a square of side int(1.4 * max(h, w)), 80 angles in [-25, 25], the angle whose
               smoothed row profile has the largest standard deviation; a second sweep of 30 angles in [-90, -50] when the
               first answer is steeper than 15 degrees.

Pinning.  scipy.ndimage.gaussian_filter1d and scipy.signal.find_peaks are the reference's own dependencies and are
imported as they are (scipy is installed here): that part is the real thing.  cv2 cannot be installed, so
getRotationMatrix2D / invertAffineTransform / warpAffine are restated from OpenCV 4.5.1 (modules/imgproc/src/imgwarp.cpp:
fixed-point source coordinates with 5 fractional bits, the float bicubic table with A = -0.75, replicated borders) [EXT]:
PARITY UNPINNED for the rotation itself -- there is no vector of the reference's to hold it against.  The 16 taps are
accumulated one by one in row-major order in float64 (OpenCV's border branch; its interior branch sums row by row);
only whether the sum is zero is used downstream (main.py:1642 `img_rotated[img_rotated != 0] = 1`).
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.signal import find_peaks

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
ROUND_DELTA = AB_SCALE // INTER_TAB_SIZE // 2


def cubic_table() -> np.ndarray:
    """float32 [32][4]: interpolateCubic(i / 32) of imgwarp.cpp (A = -0.75), evaluated in float32 like OpenCV's table."""
    A = np.float32(-0.75)
    tab = np.zeros((INTER_TAB_SIZE, 4), np.float32)
    one, two, three, four, five, eight = (np.float32(v) for v in (1, 2, 3, 4, 5, 8))
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * np.float32(1.0 / INTER_TAB_SIZE)
        c0 = ((A * (x + one) - five * A) * (x + one) + eight * A) * (x + one) - four * A
        c1 = ((A + two) * x - (A + three)) * x * x + one
        c2 = ((A + two) * (one - x) - (A + three)) * (one - x) * (one - x) + one
        c3 = one - c0 - c1 - c2
        tab[i] = (c0, c1, c2, c3)
    return tab


def rotation_matrix(center_xy, angle_deg: float) -> np.ndarray:
    """cv2.getRotationMatrix2D(center, angle, 1.0): float64 2x3, positive angle = counter-clockwise."""
    a = float(angle_deg) * np.pi / 180.0
    alpha, beta = np.cos(a), np.sin(a)
    cx, cy = float(center_xy[0]), float(center_xy[1])
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)


def invert_affine(M: np.ndarray) -> np.ndarray:
    """The in-place inversion warpAffine applies when WARP_INVERSE_MAP is not set (same operation order)."""
    m = [float(v) for v in M.reshape(-1)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return np.array(m, np.float64).reshape(2, 3)


def source_coords(Minv: np.ndarray, h: int, w: int):
    """Integer source pixel (sx, sy) and sub-pixel table indices (ax, ay) of every destination pixel (WarpAffineInvoker)."""
    m = Minv.reshape(-1)
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    adelta = np.rint(m[0] * x * AB_SCALE).astype(np.int64)
    bdelta = np.rint(m[3] * x * AB_SCALE).astype(np.int64)
    X0 = np.rint((m[1] * y + m[2]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    Y0 = np.rint((m[4] * y + m[5]) * AB_SCALE).astype(np.int64) + ROUND_DELTA
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    return sx, sy, X & (INTER_TAB_SIZE - 1), Y & (INTER_TAB_SIZE - 1)


def warp_affine_cubic_replicate(src: np.ndarray, M: np.ndarray) -> np.ndarray:
    """cv2.warpAffine(src, M, (w, h), flags=INTER_CUBIC, borderMode=BORDER_REPLICATE) for a float64 single-channel image."""
    src = np.asarray(src, np.float64)
    h, w = src.shape
    sx, sy, ax, ay = source_coords(invert_affine(M), h, w)
    tab = cubic_table()
    out = np.zeros((h, w), np.float64)
    for r in range(4):
        yy = np.clip(sy - 1 + r, 0, h - 1)
        wy = tab[ay, r]
        for c in range(4):
            xx = np.clip(sx - 1 + c, 0, w - 1)
            wgt = (wy * tab[ax, c]).astype(np.float64)      # the 2-D table entry is a float32 product
            out = out + src[yy, xx] * wgt
    return out


def rotate_image(img: np.ndarray, slope: float) -> np.ndarray:
    """main.py:159-163."""
    h, w = img.shape[:2]
    return warp_affine_cubic_replicate(img, rotation_matrix((w // 2, h // 2), slope))


def padded_square(img_patch: np.ndarray) -> np.ndarray:
    """main.py:1602-1621: the patch centred on a zero square of side int(1.4 * max(h, w))."""
    h, w = img_patch.shape[:2]
    side = int(max(h, w) * 1.4)
    sq = np.zeros((side, side), np.float64)
    cp = int(side / 2.0)
    top, left = cp - int(h / 2.0), cp - int(w / 2.0)
    sq[top:top + h, left:left + w] = img_patch
    return sq


def row_profiles(img_patch: np.ndarray, angles) -> np.ndarray:
    """int64 [len(angles)][side]: per angle, the number of non-zero pixels in every row of the rotated square."""
    sq = padded_square(img_patch)
    return np.stack([(rotate_image(sq, float(a)) != 0).sum(axis=1) for a in angles]).astype(np.int64)


def profile_statistics(y: np.ndarray, sigma: float, multiplier: float = 3.8):
    """main.py:1545-1599 on an already summed row profile y: (smoothed values at the deep minima, std of the smoothed profile).

    The reference pads the profile by 10 zeros on each side, mirrors it about its maximum, pads THAT by 10 more, smooths both
    the profile and the mirrored one with the same Gaussian, takes the maxima of each with scipy's find_peaks(height=0),
    shifts the mirrored maxima back by 20 and keeps those whose smoothed value lies below
    mean(maxima above 10) * (1 - 1/multiplier)."""
    y = np.asarray(y, np.float64)
    padded = np.concatenate([np.zeros(10), y, np.zeros(10)])
    mirrored = np.concatenate([np.zeros(10), padded.max() - padded, np.zeros(10)])
    smooth = gaussian_filter1d(y, sigma)
    valleys = find_peaks(gaussian_filter1d(mirrored, sigma), height=0)[0] - 20
    hills = find_peaks(smooth, height=0)[0]
    hill_values = smooth[hills]
    hill_values = hill_values[hill_values > 10]
    valley_values = smooth[valleys]                   # (negative indices wrap, an index >= len(y) raises -- as in the reference)
    with np.errstate(all="ignore"):
        level = np.mean(hill_values) if hill_values.size else np.float64("nan")
    limit = level - (level - 0) / multiplier
    return valley_values[valley_values < limit], np.std(smooth)


def _sweep(sq_profiles: np.ndarray, angles: np.ndarray, sigma: float) -> float:
    """One angle sweep of return_deskew_slope (main.py:1622-1667 / 1669-1716), including its bookkeeping quirk: an angle whose
    deep-minimum set is empty (mean -> NaN) is dropped from the list, and the winner's POSITION in the shortened list then
    indexes the full angle array."""
    var_res = []
    for k in range(len(angles)):
        try:
            neg, var = profile_statistics(sq_profiles[k].astype(np.float64), sigma, 20.3)
            with np.errstate(all="ignore"):
                res_me = np.mean(neg) if neg.size else np.float64("nan")
            if res_me == 0:
                res_me = 1e21
        except Exception:
            res_me, var = 1e21, 0
        if res_me != res_me:
            continue
        var_res.append(var)
    try:
        return float(angles[int(np.argmax(np.array(var_res)))])
    except Exception:
        return 0.0


def return_deskew_slope(img_patch: np.ndarray, sigma_des: float) -> float:
    """main.py:1601-1718."""
    angles = np.linspace(-25, 25, 80)
    ang = _sweep(row_profiles(img_patch, angles), angles, sigma_des)
    if abs(ang) > 15:
        angles = np.linspace(-90, -50, 30)
        ang = _sweep(row_profiles(img_patch, angles), angles, sigma_des)
    return ang
