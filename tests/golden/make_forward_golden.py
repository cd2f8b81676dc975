#!/usr/bin/env python3
"""Golden vector for the forward oracle: seeded synthetic ResNet-50-U-Net (64x64x3 -> 2 classes),
one seeded input, the oracle's fp32 softmax output.  The reference holds no golden vectors for the
forward pass (it has no tests and its network is an external .h5), so this pins the oracle against
*itself* (regression) and against torch-CPU fp64 (the generating run asserts agreement)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import keras_forward as kf  # noqa: E402
from sbb_textline_detection_amd.keras_graph import parse_model_config  # noqa: E402
from sbb_textline_detection_amd.weights import synthetic_model  # noqa: E402
from tools.synth_model import forward_torch  # noqa: E402

cfg, w = synthetic_model(2, 64, 64, seed=7)
g = parse_model_config(cfg)
x = (np.random.RandomState(11).randint(0, 256, (1, 64, 64, 3)) / 255.0).astype(np.float32)
p = kf.forward(g, w, x)
q = forward_torch(g, w, x, torch.float64)
assert np.abs(p - q).max() < 2e-3, np.abs(p - q).max()
np.savez_compressed(os.path.join(HERE, "forward_golden_64.npz"), x=x, probs=p, seed=7, classes=2)
print("max |oracle - torch f64| =", np.abs(p - q).max(), "label-1 fraction", p.argmax(-1).mean())
