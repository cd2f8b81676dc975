"""Consumers of tools/make_keras_golden.py: the forward pass pinned against REAL Keras 2.3 / TF 1.15 outputs of the reference's own
networks (main.py:58-60, 221, 287-290).  The goldens cannot be produced in this container or on the GPU box (no TF / Keras / .h5 /
network): until a maintainer runs the generator in the reference's environment, tests/golden/keras/ is empty and both tests SKIP --
which is exactly what "parity unpinned" in DESIGN.md section 5 means.  The self-check below keeps the consuming code alive meanwhile:
it writes a golden in the generator's format from the oracle itself and runs the consumers on it."""
import glob
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.environ.get("SBBSEG_KERAS_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden", "keras"))
# fp32 Keras/TF (Eigen contractions, its own summation order) against the fp32 oracle: reassociation only
TOL_ORACLE_VS_KERAS = 2e-3
EXACT_MARGIN = 2e-4


def _goldens(directory=None):
    return sorted(glob.glob(os.path.join(directory or GOLDEN_DIR, "*.golden.npz")))


def _load(npz_path):
    z = np.load(npz_path, allow_pickle=False)
    cfg = json.loads(str(z["model_config"]))
    return z, cfg


def check_oracle(npz_path, weights):
    """oracle forward == Keras' probabilities (and its named taps) within reassociation noise; labels equal away from Keras' near-ties."""
    from oracle import keras_forward as kf
    from oracle.keras_config import read_model_config
    z, cfg = _load(npz_path)
    g = read_model_config(cfg)
    x = (z["x_u8"] / 255.0).astype(np.float32)                       # main.py:239 then Keras' float32 feed
    taps = {k[5:]: None for k in z.files if k.startswith("tap__")}
    got0 = kf.forward(g, weights, x[:1], taps=taps)
    for name, val in taps.items():
        if val is None:
            continue
        ref = z["tap__" + name]
        assert val[0].shape == ref.shape, name
        assert float(np.abs(val[0] - ref).max()) <= 1e-3 * max(1.0, float(np.abs(ref).max())), f"layer {name} deviates from Keras"
    got = np.concatenate([got0] + [kf.forward(g, weights, x[k:k + 1]) for k in range(1, len(x))])
    ref = z["probs"]
    assert got.shape == ref.shape
    assert float(np.abs(got - ref).max()) <= TOL_ORACLE_VS_KERAS
    srt = np.sort(ref, axis=-1)
    decided = (srt[..., -1] - srt[..., -2]) > EXACT_MARGIN
    assert not ((got.argmax(-1) != z["labels"]) & decided).any()
    assert np.array_equal(ref.argmax(-1).astype(np.uint8), z["labels"])          # main.py:290 on Keras' own output


@pytest.mark.parametrize("npz_path", _goldens() or [None])
def test_oracle_matches_keras(npz_path):
    if npz_path is None:
        pytest.skip(f"no Keras goldens under {GOLDEN_DIR}: run tools/make_keras_golden.py where TF 1.15 + the .h5 files exist")
    from sbb_textline_detection_amd.weights import load_sbbw
    sbbw = npz_path[:-len(".golden.npz")] + ".sbbw"
    if not os.path.exists(sbbw):
        pytest.skip(f"{sbbw} missing (the generator writes it next to the .npz)")
    cfg2, weights = load_sbbw(sbbw)
    check_oracle(npz_path, weights)


@pytest.mark.gpu
@pytest.mark.parametrize("npz_path", _goldens() or [None])
def test_hip_path_matches_keras(npz_path):
    """start_new_session_and_model(<name>.sbbw) (the reference's seam, main.py:216-223) -> predict == Keras' probabilities within the
    stated tolerance of the label-exact mode; argmax == Keras' labels wherever Keras' own top-2 margin exceeds EXACT_MARGIN."""
    if npz_path is None:
        pytest.skip(f"no Keras goldens under {GOLDEN_DIR}: run tools/make_keras_golden.py where TF 1.15 + the .h5 files exist")
    from gpu_common import TOL_SOFTMAX
    from sbb_textline_detection_amd.model import start_new_session_and_model
    sbbw = npz_path[:-len(".golden.npz")] + ".sbbw"
    if not os.path.exists(sbbw):
        pytest.skip(f"{sbbw} missing")
    z, cfg = _load(npz_path)
    model, session = start_new_session_and_model(sbbw, precision="f16x3", max_batch=4)
    shp = model.layers[len(model.layers) - 1].output_shape
    assert tuple(shp[1:]) == z["probs"].shape[1:]
    got = model.predict(z["x_u8"] / 255.0)
    ref = z["probs"]
    assert float(np.abs(got - ref).max()) <= TOL_SOFTMAX["f16x3"] + TOL_ORACLE_VS_KERAS
    srt = np.sort(ref, axis=-1)
    decided = (srt[..., -1] - srt[..., -2]) > EXACT_MARGIN
    assert not ((got.argmax(-1) != z["labels"]) & decided).any()
    session.close()


def test_golden_consumer_selfcheck(tmp_path):
    """The consumer, exercised on a golden in the generator's exact format whose 'Keras outputs' come from the oracle (64x64 synthetic
    net): a format or key drift between tools/make_keras_golden.py and this file fails here, not on the one box that has TF."""
    from oracle import keras_forward as kf
    from sbb_textline_detection_amd.weights import save_sbbw, synthetic_model
    cfg, w = synthetic_model(2, 64, 64, seed=3)
    rng = np.random.Generator(np.random.PCG64(1))
    x_u8 = rng.integers(0, 256, (2, 64, 64, 3)).astype(np.uint8)
    x = (x_u8 / 255.0).astype(np.float32)
    taps = {"conv1": None, "bn_conv1": None}
    probs = np.concatenate([kf.forward_config(cfg, w, x[:1], taps=taps), kf.forward_config(cfg, w, x[1:])]).astype(np.float32)
    keys = dict(x_u8=x_u8, probs=probs, labels=probs.argmax(-1).astype(np.uint8), model_config=np.array(json.dumps(cfg)),
                keras_version=np.array("selfcheck"), tf_version=np.array("selfcheck"), numpy_version=np.array(np.__version__),
                h5_sha256=np.array("0" * 64), h5_name=np.array("selfcheck.h5"), seed=np.array(1))
    keys.update({"tap__" + k: v[0] for k, v in taps.items() if v is not None})
    npz = tmp_path / "selfcheck.golden.npz"
    np.savez_compressed(npz, **keys)
    save_sbbw(str(tmp_path / "selfcheck.sbbw"), cfg, w)
    assert _goldens(str(tmp_path)) == [str(npz)]
    check_oracle(str(npz), w)
    # the generator and this consumer agree on the key set
    src = open(os.path.join(ROOT, "tools", "make_keras_golden.py")).read()
    for k in ("x_u8", "probs", "labels", "model_config", "keras_version", "tf_version", "h5_sha256", "tap__"):
        assert k in src, k
