"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np

from oracle import keras_forward as kf
from sbb_textline_detection_amd.keras_graph import parse_model_config
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

# Stated tolerances (north_star: "within a stated float tolerance on the softmax"):
#   bf16 product path : bf16 operands/activations (8-bit mantissa) through ~60 fused layers
#   f32 check path    : fp32 everywhere, only the summation order differs from the oracle
TOL_SOFTMAX = {"bf16": 0.06, "f32": 2e-3}


def make_model(classes, h, w, seed=0, precision="bf16", max_batch=8, calib_hw=None):
    cfg, weights = calibrated_model(classes, h, w, seed=seed, calib_hw=calib_hw or min(160, max(h, w)))
    graph = parse_model_config(cfg)
    model = SegModel(cfg, weights, device=0, max_batch=max_batch, precision=precision)
    return cfg, weights, graph, model


def patches_from_page(h, w, n, seed=0):
    page = synthetic_page(max(h * 2, 600), max(w * 3, 900), seed)
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        y0 = rng.randint(0, page.shape[0] - h)
        x0 = rng.randint(0, page.shape[1] - w)
        out.append(page[y0:y0 + h, x0:x0 + w])
    return np.stack(out)


def compare_probs(ref, got, tol):
    """Returns (max abs softmax diff, label mismatches, mismatches outside the tolerance band)."""
    d = float(np.abs(ref - got).max())
    srt = np.sort(ref, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    mism = ref.argmax(-1) != got.argmax(-1)
    return d, int(mism.sum()), int((mism & (margin > 2 * tol)).sum())
