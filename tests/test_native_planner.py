"""The library's own graph reader + planner (csrc/loader.cpp, behind sbbseg_model_load) against the Python planner
(planner.py), value for value, without a GPU: both lower the same .sbbw container and every tensor, step, geometry
field and weight plane (CRC32 of the fp32 bytes) must agree -- BN folding, parity pre-sums, shortcut merge, PAIRS
stem, fused head / tail, Conv2DTranspose classes."""
import io
import json
import os
import zlib

import numpy as np
import pytest

from sbb_textline_detection_amd import _build, _capi
from sbb_textline_detection_amd.keras_graph import parse_model_config, resnet50_unet_config, transpose_unet_config
from sbb_textline_detection_amd.planner import build_plan
from sbb_textline_detection_amd.weights import save_sbbw, synthetic_weights
from tools.synth_model import calibrated_model

FIX = os.path.join(os.path.dirname(__file__), "golden", "keras23_model_config.json")


def crc(a):
    return "%08x" % (zlib.crc32(np.ascontiguousarray(a, np.float32).tobytes()) & 0xFFFFFFFF)


def python_summary(plan) -> str:
    out = [f"plan {plan.in_h} {plan.in_w} {plan.classes}"]
    for t in plan.tensors:
        out.append(f"tensor {t.H} {t.W} {t.C} {t.kind} {t.pad} {t.name}")
    for s in plan.steps:
        if s.kind == "conv":
            line = (f"conv {s.name} cout={s.cout} out={s.out_h}x{s.out_w} stride={s.out_stride[0]},{s.out_stride[1]} off={s.out_off[0]},{s.out_off[1]} "
                    f"t={s.out} relu={int(s.relu)} res={s.residual} raw={s.raw_out} head={s.head.classes if s.head else 0} macs={s.algorithmic_macs:.0f} "
                    f"scale={crc(s.scale)} shift={crc(s.shift)}")
            if s.raw_scale is not None:
                line += f" rscale={crc(s.raw_scale)} rshift={crc(s.raw_shift)}"
            if s.head is not None:
                line += f" hw={crc(s.head.w)} hs={crc(s.head.scale)} hb={crc(s.head.shift)}"
            for g in s.srcs:
                line += (f" | src t={g.tensor} ch={g.channels} k={g.kh}x{g.kw} s={g.stride_y},{g.stride_x} pad={g.pad_top},{g.pad_left} up={g.shift} "
                         f"off={g.off_y},{g.off_x} w={crc(g.w)}")
            out.append(line)
        elif s.kind == "maxpool":
            line = f"maxpool {s.name} src={s.src} dst={s.dst} k={s.k} stride={s.stride} pre={int(s.pre_relu)}"
            if s.pre_scale is not None:
                line += f" ps={crc(s.pre_scale)} pb={crc(s.pre_shift)}"
            out.append(line)
        elif s.kind == "tail":
            out.append(f"tail {s.name} src0={s.src0} img={s.img} out={s.out_h}x{s.out_w} macs={s.algorithmic_macs:.0f} w0={crc(s.w_src0)} wi={crc(s.w_img)} "
                       f"scale={crc(s.scale)} shift={crc(s.shift)} head={s.head.classes} hw={crc(s.head.w)} hs={crc(s.head.scale)} hb={crc(s.head.shift)}")
        else:
            out.append(f"head {s.name} src={s.src} cin={s.cin} classes={s.classes} hw={crc(s.w)} hs={crc(s.scale)} hb={crc(s.shift)}")
    return "\n".join(out) + "\n"


def container(cfg, w, tmp_path, name="m.sbbw"):
    path = str(tmp_path / name)
    save_sbbw(path, cfg, w)
    return open(path, "rb").read()


@pytest.fixture(scope="module", autouse=True)
def lib():
    _build.build()
    return _capi.load_library()


CASES = [("resnet50_unet 2cls 64x96 f16", lambda: calibrated_model(2, 64, 96, seed=1, calib_hw=64), "f16", 0),
         ("resnet50_unet 4cls 64x64 f16x3 (no tail)", lambda: calibrated_model(4, 64, 64, seed=2, calib_hw=64), "f16x3", 0),
         ("resnet50_unet 2cls f32 (unfused head)", lambda: calibrated_model(2, 64, 64, seed=3, calib_hw=64), "f32", 0),
         ("resnet50_unet 2cls f16, no parity split / merge", lambda: calibrated_model(2, 64, 64, seed=4, calib_hw=64), "f16", 3),
         ("resnet50_unet 2cls f16, no fused head", lambda: calibrated_model(2, 64, 64, seed=4, calib_hw=64), "f16", 4 | 8)]


@pytest.mark.parametrize("name,make,precision,flags", CASES, ids=[c[0] for c in CASES])
def test_native_planner_equals_python_planner(tmp_path, name, make, precision, flags):
    cfg, w = make()
    blob = container(cfg, w, tmp_path)
    native = _capi.native_plan_summary(blob, _capi.PRECISIONS[precision], flags)
    plan = build_plan(parse_model_config(cfg), w, parity_split=not (flags & 1), merge_shortcut=not (flags & 2),
                      fuse_head=precision != "f32" and not (flags & 4), fuse_tail=precision in ("f16", "bf16", "f16x3") and not (flags & 8))
    py = python_summary(plan)
    assert native.splitlines() == py.splitlines()


def test_native_planner_on_448_net(tmp_path):
    from sbb_textline_detection_amd.weights import synthetic_model
    cfg, w = synthetic_model(2, 448, 448, seed=0)
    blob = container(cfg, w, tmp_path)
    assert _capi.native_plan_summary(blob, _capi.PREC_F16).splitlines() == python_summary(build_plan(parse_model_config(cfg), w)).splitlines()


def test_native_planner_on_handwritten_fixture_and_transposed_decoder(tmp_path):
    cfg = json.load(open(FIX))
    w = synthetic_weights(parse_model_config(cfg), seed=5)
    blob = container(cfg, w, tmp_path, "fix.sbbw")
    assert _capi.native_plan_summary(blob, _capi.PREC_F16X3).splitlines() == \
        python_summary(build_plan(parse_model_config(cfg), w, fuse_tail=False)).splitlines()
    for k in (2, 3):
        cfg = transpose_unet_config(3, 32, 48, k=k)
        w = synthetic_weights(parse_model_config(cfg), seed=4)
        blob = container(cfg, w, tmp_path, f"t{k}.sbbw")
        assert _capi.native_plan_summary(blob, _capi.PREC_F16).splitlines() == python_summary(build_plan(parse_model_config(cfg), w)).splitlines()


def test_native_loader_errors_are_statuses(tmp_path):
    with pytest.raises(RuntimeError, match="SBBW0001"):
        _capi.native_plan_summary(b"not a container at all....", _capi.PREC_F16)
    cfg, w = calibrated_model(2, 64, 64, seed=1, calib_hw=64)
    blob = container(cfg, w, tmp_path)
    # re-pack the container with an unsupported graph (x4 upsampling): the library's own reader / planner must refuse it
    import struct
    hlen = struct.unpack("<Q", blob[8:16])[0]
    header = json.loads(blob[16:16 + hlen].decode())
    data = blob[16 + hlen + ((-(16 + hlen)) % 64):]
    for layer in header["model_config"]["config"]["layers"]:
        if layer["class_name"] == "UpSampling2D":
            layer["config"]["size"] = [4, 4]
            break
    h2 = json.dumps(header).encode()
    bad = b"SBBW0001" + struct.pack("<Q", len(h2)) + h2 + b"\0" * ((-(16 + len(h2))) % 64) + data
    with pytest.raises(RuntimeError, match="concat inputs differ|x2 upsampling"):
        _capi.native_plan_summary(bad, _capi.PREC_F16)
    with pytest.raises(RuntimeError):
        _capi.native_plan_summary(blob[:len(blob) // 2], _capi.PREC_F16)
