"""Seeded synthetic document pages (there are no sample images in the reference and no network).

``synthetic_page`` draws a white page with dark, glyph-like horizontal text bars plus mild sensor
noise -- uniform random pixels would make every pixel of a segmentation net a near-tie
(SURVEY.md 8d, config 2).  ``noise_page`` is that worst case, kept for stress tests."""
from __future__ import annotations

import numpy as np


def synthetic_page(h: int, w: int, seed: int = 0) -> np.ndarray:
    """uint8 [h, w, 3] (BGR order like cv2.imread, main.py:197 -- channels nearly equal)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    page = np.full((h, w), 235.0, np.float32)
    page += rng.normal(0, 4.0, (h, w)).astype(np.float32)
    line_h = int(rng.integers(22, 40))
    gap = int(line_h * rng.uniform(1.4, 2.2))
    left, right = int(0.08 * w), int(0.92 * w)
    y = int(0.05 * h)
    while y + line_h < 0.95 * h:
        x = left + int(rng.integers(0, 40))
        end = right - int(rng.integers(0, 0.3 * w)) if rng.random() < 0.25 else right
        while x < end:
            wl = int(rng.integers(12, 90))                   # a "word"
            x2 = min(x + wl, end)
            glyph = rng.random((line_h, x2 - x)) < 0.55     # strokes
            glyph = np.repeat(np.repeat(glyph[::3, ::3], 3, axis=0), 3, axis=1)[:line_h, :x2 - x]
            blk = page[y:y + line_h, x:x2]
            blk[glyph[:blk.shape[0], :blk.shape[1]]] = rng.uniform(20, 70)
            x = x2 + int(rng.integers(8, 22))
        y += gap
    page = np.clip(page, 0, 255)
    out = np.stack([page, page * 0.98 + 2, page * 0.96 + 4], axis=2)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def noise_page(h: int, w: int, seed: int = 0) -> np.ndarray:
    return np.random.RandomState(seed).randint(0, 256, (h, w, 3)).astype(np.uint8)
