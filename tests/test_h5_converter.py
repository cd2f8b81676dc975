"""tools/h5_to_sbbw.py round trip: a Keras-2.3-layout HDF5 (model_config attr + model_weights groups)
-> .sbbw -> identical weights.  h5py only exists in a side interpreter in this image; skipped if absent."""
import os
import subprocess

import numpy as np
import pytest

from sbb_textline_detection_amd.weights import load_sbbw, synthetic_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY_H5 = next((p for p in ("/opt/conda/bin/python3.9", "/opt/conda/bin/python") if os.path.exists(p)), None)


def _has_h5py():
    if PY_H5 is None:
        return False
    return subprocess.run([PY_H5, "-c", "import h5py, numpy"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")
@pytest.mark.parametrize("uniquify", [False, True])
def test_h5_roundtrip(tmp_path, uniquify):
    """uniquify: the file's weight_names do not start with the layer name ("conv1_1/kernel:0" inside group "conv1", nested
    datasets) -- the converter resolves them per layer group and re-keys to "<layer>/<leaf>"."""
    h5 = str(tmp_path / "model_textline_new.h5")
    tool = os.path.join(ROOT, "tools", "h5_to_sbbw.py")
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([PY_H5, tool, h5, "--fake-from-synthetic", "--classes", "4", "--size", "64", "--seed", "5"] + (["--uniquify"] if uniquify else []),
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([PY_H5, tool, h5], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    cfg, w = load_sbbw(str(tmp_path / "model_textline_new.sbbw"))
    cfg0, w0 = synthetic_model(4, 64, 64, 5)
    assert cfg == cfg0 and set(w) == set(w0)
    assert all(np.array_equal(w[k], w0[k]) for k in w0)


# ------------------------------------------------------------------------------------------------------------------
# A file the converter did NOT write: tests/golden/make_keras23_h5.py hand-builds the Keras-2.3.1 save() layout (root attrs as
# bytes, model_weights/<layer>/<scope>/<leaf>:0 nesting, weight_names / layer_names as byte-string arrays, weightless layers
# with empty weight_names) from the hand-written functional config tests/golden/keras23_model_config.json.
def _handbuilt(tmp_path, scoped):
    import json
    h5, npz = str(tmp_path / "model_textline_new.h5"), str(tmp_path / "weights.npz")
    gen = os.path.join(ROOT, "tests", "golden", "make_keras23_h5.py")
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([PY_H5, gen, h5, npz, "7"] + (["scoped"] if scoped else []), capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([PY_H5, os.path.join(ROOT, "tools", "h5_to_sbbw.py"), h5], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "keras23_model_config.json")))
    return str(tmp_path / "model_textline_new.sbbw"), dict(np.load(npz)), fix


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")
@pytest.mark.parametrize("scoped", [False, True])
def test_handbuilt_keras23_file_converts_and_plans(tmp_path, scoped):
    sbbw, want, fix = _handbuilt(tmp_path, scoped)
    cfg, w = load_sbbw(sbbw)
    assert cfg == {"class_name": fix["class_name"], "config": fix["config"]}
    assert set(w) == set(want) and all(np.array_equal(w[k], want[k]) and w[k].dtype == np.float32 for k in want)
    # the library's own reader + planner accept the converted container (no GPU needed for the plan)
    from sbb_textline_detection_amd import _capi
    txt = _capi.native_plan_summary(open(sbbw, "rb").read(), _capi.PRECISIONS["f16x3"])
    assert "conv" in txt and len(txt.splitlines()) > 5


@pytest.mark.gpu
@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py in this image")
def test_handbuilt_keras23_file_through_the_c_abi(tmp_path):
    """.h5 (hand-built Keras layout) -> tools/h5_to_sbbw.py -> start_new_session_and_model(<dir>/model_textline_new.h5) -> predict
    on the GPU == the oracle's forward of the SAME config and weights (read back from the flat npz, not from the container)."""
    from gpu_common import TOL_SOFTMAX, exact_label_check
    from oracle import keras_forward as kf
    from sbb_textline_detection_amd import clear_session
    from sbb_textline_detection_amd.model import start_new_session_and_model
    sbbw, want, fix = _handbuilt(tmp_path, False)
    rng = np.random.RandomState(3)
    x = rng.rand(5, 32, 48, 3).astype(np.float32)
    ref = kf.forward_config({"class_name": fix["class_name"], "config": fix["config"]}, want, x)
    for precision in ("f32", "f16x3"):
        model, session = start_new_session_and_model(str(tmp_path / "model_textline_new.h5"), max_batch=5, precision=precision)
        assert model.layers[-1].output_shape == (None, 32, 48, 3)                     # main.py:227-229
        got = model.predict(x)
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) < TOL_SOFTMAX[precision]
        assert exact_label_check(ref, got)[1] == 0
        session.close()
    clear_session()
