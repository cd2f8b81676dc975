#!/usr/bin/env python3
"""Hand-build an HDF5 file in the layout Keras 2.3.1 ``model.save()`` writes (what ``main.py:221`` ``load_model`` reads), from
the hand-written functional config ``tests/golden/keras23_model_config.json``.  Needs h5py (in this image:
/opt/conda/bin/python3.9); imports NOTHING from the product or its tools -- in particular not ``tools/h5_to_sbbw.py``'s own
``--fake-from-synthetic`` writer, so the converter is tested on a file it did not write.

Layout restated from keras/engine/saving.py (2.3.1) [EXT]:
  /                       attrs: keras_version = b"2.3.1", backend = b"tensorflow", model_config = JSON bytes of
                                 {"class_name": "Model", "config": {...}}   (no training_config: the reference loads with compile=False)
  /model_weights          attrs: layer_names = array of byte strings (EVERY layer, also the weightless ones),
                                 backend, keras_version
  /model_weights/<layer>  attrs: weight_names = array of byte strings "<scope>/<leaf>:0" (empty array for weightless layers)
  /model_weights/<layer>/<scope>/<leaf>:0     float32 datasets -- the weight name contains "/", so h5py nests a group named
                                              after the variable scope inside the layer's group (conv1/conv1/kernel:0)
Weight order per layer as Keras creates them: Conv2D kernel, bias; BatchNormalization gamma, beta, moving_mean, moving_variance.

    /opt/conda/bin/python3.9 tests/golden/make_keras23_h5.py out.h5 weights.npz [seed]
writes the .h5 and, beside it, the same weights as a flat npz keyed "<layer>/<leaf>:0" (what the converter must reproduce)."""
import json
import os
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def layer_weight_shapes(cfg):
    """[(layer name, [(leaf, shape)])] for every layer, in config order; channel counts follow the functional graph."""
    layers = cfg["config"]["layers"]
    channels, out = {}, []
    for l in layers:
        c, name, kind = l["config"], l["name"], l["class_name"]
        ins = [n[0] for n in l["inbound_nodes"][0]] if l["inbound_nodes"] else []
        cin = [channels[i] for i in ins]
        leaves = []
        if kind == "InputLayer":
            channels[name] = c["batch_input_shape"][-1]
        elif kind == "Conv2D":
            kh, kw = c["kernel_size"]
            leaves.append(("kernel:0", (kh, kw, cin[0], c["filters"])))
            if c.get("use_bias", True):
                leaves.append(("bias:0", (c["filters"],)))
            channels[name] = c["filters"]
        elif kind == "BatchNormalization":
            n = cin[0]
            if c.get("scale", True):
                leaves.append(("gamma:0", (n,)))
            if c.get("center", True):
                leaves.append(("beta:0", (n,)))
            leaves += [("moving_mean:0", (n,)), ("moving_variance:0", (n,))]
            channels[name] = n
        elif kind == "Concatenate":
            channels[name] = sum(cin)
        else:                                   # Activation, ZeroPadding2D, MaxPooling2D, UpSampling2D, Add, Lambda
            channels[name] = cin[0]
        out.append((name, leaves))
    return out


def main():
    out_h5, out_npz = sys.argv[1], sys.argv[2]
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    scoped = len(sys.argv) > 4 and sys.argv[4] == "scoped"      # variable scopes uniquified by TF ("conv1_1/kernel:0")
    cfg = json.load(open(os.path.join(HERE, "keras23_model_config.json")))
    rng = np.random.RandomState(seed)
    flat = {}
    with h5py.File(out_h5, "w") as f:
        f.attrs["keras_version"] = str(cfg.get("keras_version", "2.3.1")).encode("utf8")
        f.attrs["backend"] = str(cfg.get("backend", "tensorflow")).encode("utf8")
        f.attrs["model_config"] = json.dumps({"class_name": cfg["class_name"], "config": cfg["config"]}).encode("utf8")
        mw = f.create_group("model_weights")
        spec = layer_weight_shapes(cfg)
        mw.attrs["layer_names"] = np.array([name.encode("utf8") for name, _ in spec])          # dtype 'S<n>', as Keras stores it
        mw.attrs["backend"] = f.attrs["backend"]
        mw.attrs["keras_version"] = f.attrs["keras_version"]
        for name, leaves in spec:
            g = mw.create_group(name)
            scope = name + "_1" if scoped else name
            wnames = [f"{scope}/{leaf}" for leaf, _ in leaves]
            g.attrs["weight_names"] = np.array([w.encode("utf8") for w in wnames]) if wnames else np.zeros((0,), "S1")
            for (leaf, shape), wn in zip(leaves, wnames):
                if leaf == "kernel:0":
                    a = rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] * shape[1] * shape[2]))
                elif leaf in ("gamma:0", "moving_variance:0"):
                    a = rng.uniform(0.8, 1.2, shape)
                else:
                    a = rng.uniform(-0.1, 0.1, shape)
                a = a.astype(np.float32)
                g.create_dataset(wn, data=a)                     # "conv1/kernel:0" -> group conv1 inside group conv1
                flat[f"{name}/{leaf}"] = a
    np.savez(out_npz, **flat)
    print("wrote", out_h5, len(flat), "weight tensors")


if __name__ == "__main__":
    main()
