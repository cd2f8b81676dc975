#!/usr/bin/env python3
"""Pin the forward pass against the REAL reference stack: Keras 2.3 / TensorFlow 1.15 (requirements.txt:5,10) running the
three external networks of main.py:58-60 on CPU.  This script CANNOT run in the build container or on the GPU box (no TF, no
Keras, no .h5, no network): it is the one command a maintainer with the reference's environment runs once, and it turns
"parity unpinned"  into data.

    # in the reference's own environment (python 3.6/3.7, pip install -r requirements.txt), CPU is enough:
    python tools/make_keras_golden.py --models /path/to/models --out tests/golden/keras

For each of model_page_mixed_best.h5 / model_strukturerkennung.h5 / model_textline_new.h5 found under --models it

1. loads the file exactly as the reference does -- keras.models.load_model(path, compile=False)   (main.py:221),
2. draws N seeded input patches at the model's own input size, as do_prediction feeds them -- img / 255.0, float64 -> predict
   (main.py:239, 285-288); half of them document-like (dark strokes on a light page), half uniform noise (near-tie stress),
3. records label_p_pred = model.predict(patch[None]) per patch, batch 1 like the reference (main.py:287-288), as float32,
   plus np.argmax(..., axis=3) (main.py:290),
4. writes  <out>/<name>.golden.npz : x_u8 [N,H,W,3] uint8, probs [N,H,W,C] float32, labels [N,H,W] uint8, model_config (the JSON of
   the file's `model_config` root attribute, what Keras built the graph from), keras_version / tf_version / numpy_version, the
   sha256 of the .h5, and per-layer activations of patch 0 for a handful of named layers (--taps) so that a disagreement can be
   localised,
5. converts the weights with tools/h5_to_sbbw.py -> <out>/<name>.sbbw (h5py is present wherever Keras is).

What consumes it (both skip when the directory is empty -- which is the committed state):
  tests/test_keras_golden.py::test_oracle_matches_keras        (CPU)  oracle/keras_forward.py vs probs: pins the oracle
  tests/test_keras_golden.py::test_hip_path_matches_keras      (-m gpu) start_new_session_and_model(<name>.sbbw).predict vs probs,
                                                               labels vs labels wherever Keras' own top-2 margin > EXACT_MARGIN
The .npz files are small enough to commit (3 patches of 448x448: ~7 MB per model at C=2); the .sbbw files (~150 MB each) are not:
point SBBSEG_KERAS_GOLDEN_DIR at the directory instead.

Nothing of the reference's source is read or copied: the script drives Keras' public API on the reference's model files.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL_FILES = ("model_page_mixed_best.h5", "model_strukturerkennung.h5", "model_textline_new.h5")     # main.py:58-60
DEFAULT_TAPS = ("conv1", "bn_conv1", "res2c_branch2c", "res3d_branch2c", "res4f_branch2c", "res5c_branch2c")


def document_patch(rng, h, w):
    """uint8 [h, w, 3]: light page, dark glyph-like strokes in text lines, mild noise (the regime the nets were trained on)."""
    page = np.full((h, w), 232.0) + rng.normal(0, 4.0, (h, w))
    line_h = int(rng.integers(18, 34))
    y = int(rng.integers(4, 30))
    while y + line_h < h - 4:
        x = int(rng.integers(6, 40))
        while x < w - 10:
            x2 = min(x + int(rng.integers(10, 80)), w - 6)
            strokes = rng.random((line_h, x2 - x)) < 0.5
            page[y:y + line_h, x:x2][strokes] = rng.uniform(20, 80)
            x = x2 + int(rng.integers(6, 20))
        y += int(line_h * rng.uniform(1.4, 2.2))
    page = np.clip(page, 0, 255)
    return np.clip(np.rint(np.stack([page, page * 0.98 + 2, page * 0.96 + 4], axis=2)), 0, 255).astype(np.uint8)


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def model_config_of(h5_path):
    import h5py
    with h5py.File(h5_path, "r") as f:
        raw = f.attrs["model_config"]
    return raw.decode("utf-8") if isinstance(raw, bytes) else str(raw)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--models", required=True, help="directory holding the reference's .h5 files (README.md:42)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "keras"))
    ap.add_argument("--patches", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--taps", default=",".join(DEFAULT_TAPS), help="layer names whose output for patch 0 is stored too (missing ones are skipped)")
    ap.add_argument("--no-sbbw", action="store_true", help="do not convert the weights (only the .npz)")
    args = ap.parse_args()

    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")          # the reference's CPU path
    os.environ.setdefault("TF_CPP_MIN_LOG_LEVEL", "3")
    import keras
    import tensorflow as tf
    from keras import backend as K
    from keras.models import Model, load_model

    os.makedirs(args.out, exist_ok=True)
    found = 0
    for fname in MODEL_FILES:
        path = os.path.join(args.models, fname)
        if not os.path.exists(path):
            print(f"skip {fname}: not under {args.models}")
            continue
        found += 1
        name = os.path.splitext(fname)[0]
        K.clear_session()                                       # main.py:2065
        model = load_model(path, compile=False)                 # main.py:221
        _, H, W, C = model.layers[len(model.layers) - 1].output_shape           # main.py:227-229
        rng = np.random.Generator(np.random.PCG64(args.seed + found))
        xs = []
        for k in range(args.patches):
            xs.append(document_patch(rng, H, W) if k % 2 == 0 else rng.integers(0, 256, (H, W, 3)).astype(np.uint8))
        x_u8 = np.stack(xs)
        probs = np.stack([model.predict((x / float(255.0)).reshape(1, H, W, 3))[0] for x in x_u8]).astype(np.float32)    # main.py:239, 287-288
        labels = np.argmax(probs, axis=3).astype(np.uint8)      # main.py:290
        taps = {}
        for lname in [t for t in args.taps.split(",") if t]:
            try:
                sub = Model(inputs=model.input, outputs=model.get_layer(lname).output)
            except ValueError:
                continue
            taps["tap__" + lname] = sub.predict((x_u8[0] / float(255.0)).reshape(1, H, W, 3))[0].astype(np.float32)
        out_npz = os.path.join(args.out, name + ".golden.npz")
        np.savez_compressed(out_npz, x_u8=x_u8, probs=probs, labels=labels, model_config=np.array(model_config_of(path)),
                            keras_version=np.array(keras.__version__), tf_version=np.array(tf.__version__),
                            numpy_version=np.array(np.__version__), h5_sha256=np.array(sha256_of(path)), h5_name=np.array(fname),
                            seed=np.array(args.seed + found), **taps)
        print(f"wrote {out_npz}: {args.patches} patches of {H}x{W}, {C} classes, taps {sorted(k[5:] for k in taps)}")
        if not args.no_sbbw:
            sys.path.insert(0, ROOT)
            from tools.h5_to_sbbw import h5_to_sbbw
            h5_to_sbbw(path, os.path.join(args.out, name + ".sbbw"))
    if not found:
        raise SystemExit(f"none of {MODEL_FILES} under {args.models}")


if __name__ == "__main__":
    main()
