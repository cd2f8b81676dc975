"""A hand-written Keras-2.3 ``model_config`` (tests/golden/keras23_model_config.json: functional ``Model``, marshalled
``Lambda`` placeholders, the three ZeroPadding2D tuple forms, BN ``axis`` as int and as list, a projection-shortcut
block, one_side_pad on a skip, the raw conv1 skip, Concatenate axis 3 / -1) read by TWO independently written readers:
the product's ``keras_graph.parse_model_config`` and the oracle's ``oracle/keras_config.read_model_config``.
Shapes are also pinned against numbers worked out by hand, so both readers being wrong the same way still fails."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from oracle import keras_forward as kf
from oracle.keras_config import read_model_config
from plan_interp import run_plan
from sbb_textline_detection_amd.keras_graph import parse_model_config, resnet50_unet_config
from sbb_textline_detection_amd.planner import build_plan
from sbb_textline_detection_amd.weights import synthetic_weights
from tools.synth_model import forward_torch

FIX = os.path.join(os.path.dirname(__file__), "golden", "keras23_model_config.json")

# worked by hand from the layer configs: 32x48x3 -> pad 3 -> 7x7 s2 valid: (38-7)//2+1 = 16, (54-7)//2+1 = 24 -> pool 3/2
# valid: 7 x 11 -> one_side_pad: 8 x 12 -> 1x1 stride-2 valid: 4 x 6 -> x2: 8 x 12 -> ... -> 32 x 48 x 3
HAND_SHAPES = {"zero_padding2d_1": (38, 54, 3), "conv1": (16, 24, 16), "max_pooling2d_1": (7, 11, 16), "res2a_branch2b": (7, 11, 8),
               "add_1": (7, 11, 32), "add_2": (7, 11, 32), "zero_padding2d_2": (9, 13, 32), "lambda_1": (8, 12, 32),
               "conv2d_1": (4, 6, 24), "up_sampling2d_1": (8, 12, 24), "concatenate_1": (8, 12, 56), "zero_padding2d_3": (10, 14, 56),
               "conv2d_2": (8, 12, 16), "concatenate_2": (16, 24, 32), "zero_padding2d_4": (18, 26, 32), "conv2d_3": (16, 24, 16),
               "concatenate_3": (32, 48, 19), "conv2d_4": (32, 48, 32), "activation_11": (32, 48, 3)}
HAND_PADS = {"zero_padding2d_1": (3, 3, 3, 3), "zero_padding2d_2": (1, 1, 1, 1), "zero_padding2d_3": (1, 1, 1, 1),
             "zero_padding2d_4": (1, 1, 1, 1), "zero_padding2d_5": (1, 1, 1, 1)}


@pytest.fixture(scope="module")
def cfg():
    return json.load(open(FIX))


def _same(a, b):
    assert len(a.nodes) == len(b.nodes)
    for x, y in zip(a.nodes, b.nodes):
        assert (x.name, x.op, tuple(x.inputs), tuple(x.out_shape)) == (y.name, y.op, tuple(y.inputs), tuple(y.out_shape)), (x, y)
        for k, v in x.attrs.items():
            assert y.attrs[k] == v, (x.name, k, v, y.attrs[k])
    assert (a.input_name, a.output_name) == (b.input_name, b.output_name)


def test_two_readers_agree_and_match_hand_shapes(cfg):
    a, b = read_model_config(cfg), parse_model_config(cfg)
    _same(a, b)
    byn = a.by_name()
    for name, shp in HAND_SHAPES.items():
        assert tuple(byn[name].out_shape) == shp, name
    for name, pad in HAND_PADS.items():
        assert byn[name].attrs["pad"] == pad, name
    assert byn["lambda_1"].op == "crop_last" and byn["conv2d_1"].attrs["strides"] == (2, 2)
    assert byn["concatenate_2"].inputs[1] == "conv1"            # the skip is taken BEFORE bn_conv1
    assert a.output_shape == (32, 48, 3) and byn["activation_11"].output_shape == (None, 32, 48, 3)   # main.py:227-229


@pytest.mark.parametrize("classes,hw", [(2, (448, 448)), (4, (224, 320))])
def test_two_readers_agree_on_generated_resnet50_unet(classes, hw):
    c = resnet50_unet_config(classes, *hw)
    _same(read_model_config(c), parse_model_config(c))
    _same(read_model_config(json.dumps(c)), parse_model_config(json.dumps(c)))        # JSON text, as stored in the .h5 attribute


def test_forward_three_ways_on_the_fixture(cfg):
    """oracle (own reader + C conv), torch-CPU (product parser + torch ops) and the product's fused plan (numpy
    interpreter of planner.build_plan) agree on the fixture -- with and without the decoder parity split."""
    g_prod = parse_model_config(cfg)
    w = synthetic_weights(g_prod, seed=5)
    rng = np.random.RandomState(2)
    for name in list(w):                                         # BN statistics that keep activations O(1)
        if name.endswith("moving_variance:0"):
            w[name] = rng.uniform(0.5, 2.0, w[name].shape).astype(np.float32)
    x = rng.rand(2, 32, 48, 3).astype(np.float32)
    p = kf.forward_config(cfg, w, x)
    q = forward_torch(g_prod, w, x, torch.float64)
    assert p.shape == (2, 32, 48, 3) and np.abs(p - q).max() < 1e-4
    for parity in (True, False):
        lab, pr, _ = run_plan(build_plan(g_prod, w, parity_split=parity), x)
        assert np.abs(pr - p).max() < 5e-4
        srt = np.sort(p, axis=-1)
        decided = (srt[..., -1] - srt[..., -2]) > 2e-3
        assert np.array_equal(lab[decided], p.argmax(-1)[decided])


def test_unrecognised_lambda_is_refused_by_both_readers(cfg):
    bad = copy.deepcopy(cfg)
    for layer in bad["config"]["layers"]:
        if layer["name"] == "zero_padding2d_2":
            layer["config"]["padding"] = [[0, 2], [0, 2]]        # not one_side_pad's (1, 1) padding in front of the Lambda
    for reader in (read_model_config, parse_model_config):
        with pytest.raises(ValueError):
            reader(bad)


def test_batchnorm_without_scale_or_center(cfg):
    """BatchNormalization(scale=False) has no gamma, (center=False) no beta (Keras build()): weight_specs, the planner, the
    oracle and the torch cross-check all honour the flags."""
    c = copy.deepcopy(cfg)
    for layer in c["config"]["layers"]:
        if layer["name"] == "bn2a_branch2a":
            layer["config"]["scale"] = False
        if layer["name"] == "batch_normalization_2":
            layer["config"]["center"] = False
    g = parse_model_config(c)
    names = [n for n, _ in g.weight_specs()]
    assert "bn2a_branch2a/gamma:0" not in names and "bn2a_branch2a/beta:0" in names
    assert "batch_normalization_2/beta:0" not in names and "batch_normalization_2/gamma:0" in names
    w = synthetic_weights(g, seed=6)
    x = np.random.RandomState(3).rand(1, 32, 48, 3).astype(np.float32)
    p = kf.forward_config(c, w, x)
    q = forward_torch(g, w, x, torch.float64)
    lab, pr, _ = run_plan(build_plan(g, w), x)
    assert np.abs(p - q).max() < 1e-4 and np.abs(pr - p).max() < 5e-4


def test_lambda_lowering_warns_and_can_be_refused(cfg, monkeypatch):
    """A Lambda's body is opaque bytecode: the parser lowers the one after ZeroPadding2D((1,1)) as upstream's one_side_pad crop, says
    so in a warning that names the layer, refuses named functions / bound arguments, and refuses everything in strict mode."""
    import warnings
    from sbb_textline_detection_amd import keras_graph
    keras_graph._LAMBDA_WARNED.clear()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        parse_model_config(cfg)
        parse_model_config(cfg)
    hits = [w for w in rec if "lambda_1" in str(w.message) and "one_side_pad" in str(w.message)]
    assert len(hits) == 1                                   # announced, and only once per process and layer
    bad = copy.deepcopy(cfg)
    lam = next(l for l in bad["config"]["layers"] if l["class_name"] == "Lambda")
    lam["config"]["function_type"] = "function"
    with pytest.raises(ValueError, match="not the one_side_pad crop"):
        parse_model_config(bad)
    lam["config"]["function_type"] = "lambda"
    lam["config"]["arguments"] = {"k": 1}
    with pytest.raises(ValueError, match="not the one_side_pad crop"):
        parse_model_config(bad)
    monkeypatch.setenv("SBBSEG_STRICT_LAMBDA", "1")
    with pytest.raises(ValueError, match="SBBSEG_STRICT_LAMBDA"):
        parse_model_config(cfg)
