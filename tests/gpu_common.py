"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np

from oracle import keras_forward as kf
from oracle.keras_config import read_model_config
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

# Stated tolerances (north_star: "within a stated float tolerance on the softmax"), measured on the
# seeded random-weight net -- a noise amplifier: every layer's rounding is carried, undamped, through
# ~60 fused layers, and |logit0-logit1| is unimodal around 0, so this is the worst case for labels:
#   f32  check path   : fp32 everywhere, only summation order differs          -> 2e-3  (measured 5e-5)
#   f16  fast path    : 11-bit significands, fp32 accumulate/epilogue           -> 0.115 (measured 0.084-0.095)
# Labels must agree wherever the oracle's top-2 margin exceeds 2 x tolerance.
#   f16x3 label-exact : split fp16 (hi + lo, ~22 bits), 3 MFMAs per product       -> 1e-3  (measured 9e-5 at 448x448)
# (f16: measured 0.084-0.095 on the 448x448 seeded nets, r02 -> measured + 20 %; bf16 is an A/B mode only -- 8-bit
#  significands through ~60 layers give 0.34 / 7 % labels -- and carries no softmax tolerance: a bound of 0.6 on a
#  quantity in [0, 1] would assert nothing)
TOL_SOFTMAX = {"f16": 0.115, "f32": 2e-3, "f16x3": 1e-3}
# fp16 fast mode: measured label mismatch fractions 1.9 % (2 classes) / 2.8 % (4 classes) on the noise-like seeded nets
TOL_LABEL_FRAC_F16 = 0.035
# max |err| / max |ref| per fused layer output
TOL_LAYER_REL = {"bf16": 0.25, "f16": 0.04, "f32": 2e-4, "f16x3": 2e-4}
# label-exact modes: labels must equal the oracle's wherever its top-2 softmax margin exceeds this (measured worst margin
# among differing pixels: 7e-5, the fp32 oracle's own reassociation noise; tightened from 1e-3 in round 3)
EXACT_MARGIN = 2e-4


def make_model(classes, h, w, seed=0, precision="f16", max_batch=8, calib_hw=None, decisive=False):
    cfg, weights = calibrated_model(classes, h, w, seed=seed, calib_hw=calib_hw or min(160, max(h, w)), decisive=decisive)
    graph = read_model_config(cfg)          # the ORACLE's own reader of the model_config (not the product's parser)
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


def exact_label_check(ref, got, margin=EXACT_MARGIN):
    """(label mismatches, mismatches where the oracle's top-2 margin exceeds `margin`) -- the second must be 0
    in a label-exact mode."""
    srt = np.sort(ref, axis=-1)
    m = srt[..., -1] - srt[..., -2]
    mism = ref.argmax(-1) != got.argmax(-1)
    return int(mism.sum()), int((mism & (m > margin)).sum())


def compare_probs(ref, got, tol):
    """Returns (max abs softmax diff, label mismatches, mismatches outside the tolerance band)."""
    d = float(np.abs(ref - got).max())
    srt = np.sort(ref, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    mism = ref.argmax(-1) != got.argmax(-1)
    return d, int(mism.sum()), int((mism & (margin > 2 * tol)).sum())
