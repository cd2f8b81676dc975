"""oracle/keras_forward.py: golden regression + independent torch-CPU cross-check (fp32 and fp64).
Forward parity vs the real Keras/TF path is UNPINNED (no .h5, no TF, the reference has no vectors)."""
import os

import numpy as np
import pytest
import torch

from oracle import keras_forward as kf
from oracle.keras_config import read_model_config
from sbb_textline_detection_amd.keras_graph import parse_model_config, resnet50_unet_config
from sbb_textline_detection_amd.weights import synthetic_model
from tools.synth_model import calibrated_model, forward_torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "forward_golden_64.npz")


def test_forward_golden_regression():
    d = np.load(GOLD)
    cfg, w = synthetic_model(int(d["classes"]), 64, 64, seed=int(d["seed"]))
    p = kf.forward(read_model_config(cfg), w, d["x"])
    assert p.shape == d["probs"].shape and p.dtype == np.float32
    assert np.abs(p - d["probs"]).max() < 1e-5
    assert np.array_equal(p.argmax(-1), d["probs"].argmax(-1))


@pytest.mark.parametrize("classes,hw", [(2, (64, 96)), (4, (96, 64))])
def test_oracle_vs_torch_cpu(classes, hw):
    cfg, w = calibrated_model(classes, hw[0], hw[1], seed=3, calib_hw=64)
    g = parse_model_config(cfg)
    x = (np.random.RandomState(5).randint(0, 256, (2, hw[0], hw[1], 3)) / 255.0).astype(np.float32)
    p = kf.forward(read_model_config(cfg), w, x)          # oracle: its own reader + C conv; torch: the product's parser + torch ops
    q32 = forward_torch(g, w, x, torch.float32)
    q64 = forward_torch(g, w, x, torch.float64)
    assert np.abs(p - q64).max() < 1e-3 and np.abs(p - q32).max() < 1e-3
    margin = np.sort(q64, axis=-1)
    decided = (margin[..., -1] - margin[..., -2]) > 1e-3
    assert np.array_equal(p.argmax(-1)[decided], q64.argmax(-1)[decided])
    assert np.allclose(p.sum(-1), 1.0, atol=1e-5)


def test_architecture_numbers_match_survey():
    g = parse_model_config(resnet50_unet_config(2, 448, 448))
    byn = g.by_name()
    assert g.output_shape == (448, 448, 2)
    assert byn["conv1"].out_shape == (224, 224, 64)
    assert byn["max_pooling2d_1"].out_shape == (111, 111, 64)          # valid 3x3 s2 pooling
    assert byn["res3a_branch2a"].out_shape == (56, 56, 128)
    assert byn["lambda_1"].out_shape == (112, 112, 256)                # one_side_pad
    n_params = sum(int(np.prod(s)) for name, s in g.weight_specs() if name.endswith("kernel:0"))
    assert abs(n_params - 38.07e6) < 0.05e6


def test_conv_same_padding_matches_tf_rule():
    # even input, stride 2, 3x3 'same' -> TF pads (0 top/left, 1 bottom/right)
    x = np.arange(1 * 4 * 4 * 1, dtype=np.float32).reshape(1, 4, 4, 1)
    w = np.ones((3, 3, 1, 1), np.float32)
    y = kf.conv2d(x, w, None, (2, 2), "same")
    assert y.shape == (1, 2, 2, 1)
    assert y[0, 0, 0, 0] == x[0, 0:3, 0:3, 0].sum() and y[0, 1, 1, 0] == x[0, 2:4, 2:4, 0].sum()
