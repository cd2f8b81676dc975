"""Conv2DTranspose decoders (BASELINE.json north_star: "the decoder's transposed-conv upsamples"; SURVEY.md 0.5: support
both UpSampling2D and Conv2DTranspose): the oracle's op against torch, and the planner's output-parity lowering
(k = 2: four 1x1 convs; k = 3: 2x2 / 2x1 / 1x2 / 1x1 sub-kernels, ...) against the unfused oracle.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import keras_forward as kf
from oracle.keras_config import read_model_config
from plan_interp import run_plan
from sbb_textline_detection_amd.keras_graph import parse_model_config, transpose_unet_config
from sbb_textline_detection_amd.planner import build_plan
from sbb_textline_detection_amd.weights import synthetic_weights
from tools.synth_model import forward_torch


@pytest.mark.parametrize("k,padding", [(2, "same"), (2, "valid"), (3, "same"), (3, "valid"), (4, "same"), (5, "valid")])
def test_oracle_conv2d_transpose_vs_torch(k, padding):
    rng = np.random.RandomState(k)
    x = rng.randn(2, 5, 7, 6).astype(np.float32)
    w = rng.randn(k, k, 4, 6).astype(np.float32)                       # Keras layout [kh][kw][out][in]
    b = rng.randn(4).astype(np.float32)
    y = kf.conv2d_transpose(x, w, b, (2, 2), padding)
    full = F.conv_transpose2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w).permute(3, 2, 0, 1).double(),
                              torch.from_numpy(b).double(), stride=2).permute(0, 2, 3, 1).numpy()     # (H-1)*2 + k rows
    if padding == "same":
        p = max(k - 2, 0) // 2
        ref = full[:, p:p + 10, p:p + 14]
        assert y.shape == (2, 10, 14, 4)
    else:
        ref = full
        assert y.shape == (2, 10 + max(k - 2, 0), 14 + max(k - 2, 0), 4)  # Keras: H*s + max(k - s, 0)
    assert np.abs(y - ref).max() < 1e-4


@pytest.mark.parametrize("k,padding", [(2, "same"), (3, "same"), (2, "valid"), (4, "same")])
def test_transpose_decoder_lowering_equals_oracle(k, padding):
    cfg = transpose_unet_config(3, 32, 48, k=k, padding=padding)
    g, g_oracle = parse_model_config(cfg), read_model_config(cfg)
    assert [(n.name, n.op, tuple(n.out_shape)) for n in g.nodes] == [(n.name, n.op, tuple(n.out_shape)) for n in g_oracle.nodes]
    w = synthetic_weights(g, 4)
    x = np.random.RandomState(1).rand(2, 32, 48, 3).astype(np.float32)
    p = kf.forward(g_oracle, w, x)
    q = forward_torch(g, w, x, torch.float64)
    assert np.abs(p - q).max() < 1e-4
    plan = build_plan(g, w)
    classes = [s for s in plan.steps if s.kind == "conv" and ":t" in s.name]
    assert len(classes) == 8 and all(s.out_stride == (2, 2) for s in classes)
    if k == 2:
        assert all((s.srcs[0].kh, s.srcs[0].kw) == (1, 1) for s in classes)
    if k == 3:
        assert sorted((s.srcs[0].kh, s.srcs[0].kw) for s in classes[:4]) == [(1, 1), (1, 2), (2, 1), (2, 2)]
    assert plan.executed_macs_per_patch() == plan.macs_per_patch()          # a scatter has no redundant taps to pre-sum
    lab, pr, _ = run_plan(plan, x)
    assert np.abs(pr - p).max() < 5e-4
    srt = np.sort(p, axis=-1)
    decided = (srt[..., -1] - srt[..., -2]) > 2e-3
    assert np.array_equal(lab[decided], p.argmax(-1)[decided])
