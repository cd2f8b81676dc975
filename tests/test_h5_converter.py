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
