"""planner.py lowering (fusions, PAIRS stem, one_side_pad offset, head) checked on CPU against the
unfused oracle through tests/plan_interp.py -- no GPU involved."""
import numpy as np
import pytest

from oracle import keras_forward as kf
from oracle.keras_config import read_model_config
from plan_interp import run_plan
from sbb_textline_detection_amd.keras_graph import parse_model_config, resnet50_unet_config
from sbb_textline_detection_amd.planner import PlanError, build_plan
from sbb_textline_detection_amd.synthetic import synthetic_page
from sbb_textline_detection_amd.weights import synthetic_model, synthetic_weights
from tools.synth_model import calibrated_model


@pytest.mark.parametrize("classes,hw,parity,fuse", [(2, (96, 128), True, True), (4, (64, 64), True, False),
                                                     (2, (64, 96), False, True), (4, (64, 64), False, False)])
def test_plan_equals_oracle(classes, hw, parity, fuse):
    cfg, w = calibrated_model(classes, hw[0], hw[1], seed=1, calib_hw=64)
    g = parse_model_config(cfg)
    plan = build_plan(g, w, parity_split=parity, fuse_head=fuse, fuse_tail=(classes == 2))
    page = synthetic_page(300, 400, 5)
    x = np.stack([page[10:10 + hw[0], 20:20 + hw[1]], page[100:100 + hw[0], 200:200 + hw[1]]]).astype(np.float32) / 255
    ref = kf.forward(read_model_config(cfg), w, x)
    lab, pr, _ = run_plan(plan, x)
    assert np.abs(ref - pr).max() < 5e-4
    srt = np.sort(ref, axis=-1)
    decided = (srt[..., -1] - srt[..., -2]) > 2e-3
    assert np.array_equal(lab[decided], ref.argmax(-1)[decided])


def test_plan_structure_448():
    cfg, w = synthetic_model(2, 448, 448, 0)
    plan = build_plan(parse_model_config(cfg), w, parity_split=False, fuse_head=False, merge_shortcut=False)
    kinds = [s.kind for s in plan.steps]
    assert kinds.count("conv") == 59 and kinds.count("maxpool") == 1 and kinds[-1] == "head"
    assert plan.macs_per_patch() == 47418195968                       # 47.418 GMAC
    assert plan.executed_macs_per_patch() > plan.macs_per_patch()      # only the stem's zero taps/channels
    stem = plan.steps[0].srcs[0]
    assert (stem.kh, stem.kw, stem.stride_y, stem.stride_x) == (7, 4, 2, 1)
    # the stem stores only its pre-BN tensor (the f1 skip); bn_conv1 + relu ride in the max-pool
    assert plan.steps[0].raw_out == -1 and not plan.steps[0].relu and plan.steps[1].kind == "maxpool" and plan.steps[1].pre_relu
    dec = {s.name: s for s in plan.steps if s.kind == "conv" and len(s.srcs) == 2}
    assert len(dec) == 5 and all(s.srcs[0].shift == 1 for s in dec.values())
    f2 = [s for s in dec.values() if s.srcs[1].off_y == 1]
    assert len(f2) == 1 and f2[0].srcs[1].off_x == 1                   # one_side_pad became an offset
    assert sum(1 for s in plan.steps if s.kind == "conv" and s.residual >= 0) == 16
    # projection-shortcut blocks merged: shortcut conv + conv c + Add = one two-source conv, BN scales in the weights
    merged = build_plan(parse_model_config(cfg), w, parity_split=False, fuse_head=False)
    convs = [s for s in merged.steps if s.kind == "conv"]
    assert len(convs) == 55 and sum(1 for s in convs if s.residual >= 0) == 12
    two = [s for s in convs if len(s.srcs) == 2 and not s.srcs[0].shift]
    assert len(two) == 4 and all(np.all(s.scale == 1.0) and s.relu for s in two)
    assert [(s.srcs[0].channels, s.srcs[1].channels, s.cout, s.srcs[1].stride_y) for s in two] == \
        [(64, 64, 256, 1), (128, 256, 512, 2), (256, 512, 1024, 2), (512, 1024, 2048, 2)]
    assert merged.macs_per_patch() == 47418195968


def test_fused_tail_448():
    cfg, w = synthetic_model(4, 448, 448, 0)
    plan = build_plan(parse_model_config(cfg), w)
    kinds = [s.kind for s in plan.steps]
    assert kinds.count("tail") == 1 and kinds[-1] == "tail" and "head" not in kinds
    assert kinds.count("conv") == 50 + 4 * 4                              # dec1-dec4 parity classes; dec5 lives in the tail
    assert plan.macs_per_patch() == 47418195968 + 448 * 448 * 32 * 2     # (4 classes instead of 2 in the head)
    tail = plan.steps[-1]
    assert tail.w_src0.shape == (3, 3, 64, 32) and tail.w_img.shape == (3, 3, 3, 32)


def test_parity_split_and_fused_head_448():
    cfg, w = synthetic_model(2, 448, 448, 0)
    plan = build_plan(parse_model_config(cfg), w, fuse_tail=False)
    kinds = [s.kind for s in plan.steps]
    assert kinds.count("conv") == 50 + 5 * 4 and "head" not in kinds    # 5 decoder convs x 4 parity classes
    assert plan.macs_per_patch() == 47418195968                       # algorithmic work is unchanged ...
    assert plan.executed_macs_per_patch() < 0.80 * plan.macs_per_patch()   # ... but > 20 % fewer MACs are issued
    par = [s for s in plan.steps if s.kind == "conv" and s.out_stride == (2, 2)]
    assert len(par) == 20 and all((s.srcs[0].kh, s.srcs[0].kw, s.srcs[0].shift) == (2, 2, 0) for s in par)
    assert all((s.srcs[1].kh, s.srcs[1].stride_y) == (3, 2) for s in par)
    heads = [s for s in par if s.head is not None]
    assert len(heads) == 4 and all(s.out == -1 and s.cout == 32 for s in heads)
    assert sum(1 for t in plan.tensors if t.kind == "unused") == 1


def test_unsupported_graph_raises_plan_error():
    cfg = resnet50_unet_config(2, 64, 64)
    for layer in cfg["config"]["layers"]:
        if layer["class_name"] == "UpSampling2D":
            layer["config"]["size"] = [4, 4]
            break
    with pytest.raises((PlanError, ValueError)):
        g = parse_model_config(cfg)
        build_plan(g, synthetic_weights(g, 0))
