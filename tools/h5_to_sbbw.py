#!/usr/bin/env python3
"""Offline converter: Keras-2.3 HDF5 model (what the reference loads, main.py:58-60, 221) -> .sbbw.

    python tools/h5_to_sbbw.py  models/model_textline_new.h5            # writes models/model_textline_new.sbbw

Needs h5py (any interpreter that has it -- in this image: /opt/conda/bin/python3.9); inference never
does.  Reads the root attribute ``model_config`` (JSON) and ``model_weights/<layer>/<weight_name>``
datasets in each layer's ``weight_names`` order -- nothing is executed, Lambda layers stay opaque
(the graph parser pattern-matches them).  ``--fake-from-synthetic`` writes a Keras-layout .h5 from
the seeded synthetic model instead (used by the round-trip test; no real .h5 exists offline).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _load_pkg():
    # import the two pure-numpy modules without pulling the package __init__ (ctypes lib, etc.)
    import importlib.util
    mods = {}
    for name in ("keras_graph", "weights"):
        spec = importlib.util.spec_from_file_location(f"sbb_textline_detection_amd.{name}",
                                                      os.path.join(ROOT, "sbb_textline_detection_amd", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        mods[name] = (spec, m)
    import types
    pkg = types.ModuleType("sbb_textline_detection_amd")
    pkg.__path__ = [os.path.join(ROOT, "sbb_textline_detection_amd")]
    sys.modules.setdefault("sbb_textline_detection_amd", pkg)
    for name in ("keras_graph", "weights"):
        spec, m = mods[name]
        spec.loader.exec_module(m)
    return mods["keras_graph"][1], mods["weights"][1]


def _s(x):
    return x.decode("utf-8") if isinstance(x, (bytes, np.bytes_)) else str(x)


def h5_to_sbbw(h5_path: str, out_path: str) -> None:
    import h5py
    kg, wt = _load_pkg()
    with h5py.File(h5_path, "r") as f:
        cfg = json.loads(_s(f.attrs["model_config"]))
        grp = f["model_weights"] if "model_weights" in f else f
        graph = kg.parse_model_config(cfg)
        # Keras keys a layer's group by the LAYER name, but the weight_names inside carry the variable scope, which is often
        # uniquified ("conv1_1/kernel:0", "batch_normalization_1_1/gamma:0") and so need not start with the layer name.  Resolve
        # per layer: by leaf name ("kernel:0") when unique, else by position in weight_names (Keras' creation order = ours),
        # and re-key to "<layer name>/<leaf>" -- what the planner and the oracle look up.
        expected = {}
        for name, shape in graph.weight_specs():
            expected.setdefault(name.split("/")[0], []).append((name.split("/", 1)[1], tuple(shape)))
        weights, problems = {}, []
        for layer, leaves in expected.items():
            if layer not in grp:
                problems.append(f"{layer}: no group in model_weights")
                continue
            g = grp[layer]
            names = [_s(w) for w in g.attrs["weight_names"]]
            for pos, (leaf, shape) in enumerate(leaves):
                byleaf = [w for w in names if w.rsplit("/", 1)[-1] == leaf]
                pick = byleaf[0] if len(byleaf) == 1 else (names[pos] if pos < len(names) and len(names) == len(leaves) else None)
                if pick is None:
                    problems.append(f"{layer}/{leaf}: not found among {names}")
                    continue
                arr = np.asarray(g[pick], np.float32)
                if tuple(arr.shape) != shape:
                    problems.append(f"{layer}/{leaf}: shape {arr.shape} != expected {shape} (from {pick})")
                    continue
                weights[f"{layer}/{leaf}"] = arr
    if problems:
        raise SystemExit(f"{h5_path}: cannot resolve weights:\n  " + "\n  ".join(problems[:12]))
    wt.save_sbbw(out_path, cfg, weights)
    print(f"wrote {out_path}: {len(graph.nodes)} layers, {sum(v.size for v in weights.values())/1e6:.2f} M parameters, "
          f"input {graph.input_shape}, output {graph.output_shape}")


def fake_h5(out_path: str, classes: int, size: int, seed: int, uniquify: bool = False) -> None:
    """A Keras-2.3-layout .h5 of the seeded synthetic model.  ``uniquify``: weight_names carry a uniquified variable scope
    ("<layer>_1/kernel:0") and nested datasets, like files written after the layer names were taken once already."""
    import h5py
    kg, wt = _load_pkg()
    cfg, weights = wt.synthetic_model(classes, size, size, seed)
    graph = kg.parse_model_config(cfg)
    with h5py.File(out_path, "w") as f:
        f.attrs["model_config"] = json.dumps(cfg).encode("utf-8")
        f.attrs["keras_version"] = b"2.3.1"
        f.attrs["backend"] = b"tensorflow"
        mw = f.create_group("model_weights")
        mw.attrs["layer_names"] = [n.name.encode() for n in graph.nodes]
        per_layer = {}
        for name, _shape in graph.weight_specs():
            per_layer.setdefault(name.split("/")[0], []).append(name)
        for n in graph.nodes:
            g = mw.create_group(n.name)
            names = per_layer.get(n.name, [])
            stored = [(f"{n.name}_1/{w.split('/', 1)[1]}" if uniquify else w) for w in names]
            g.attrs["weight_names"] = [w.encode() for w in stored]
            for w, sw in zip(names, stored):
                g.create_dataset(sw, data=weights[w])                  # (a path with "/" creates the nested scope group, as Keras does)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("h5")
    ap.add_argument("-o", "--out")
    ap.add_argument("--fake-from-synthetic", action="store_true")
    ap.add_argument("--uniquify", action="store_true", help="with --fake-from-synthetic: uniquified weight scopes in weight_names")
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.fake_from_synthetic:
        fake_h5(a.h5, a.classes, a.size, a.seed, a.uniquify)
    else:
        h5_to_sbbw(a.h5, a.out or os.path.splitext(a.h5)[0] + ".sbbw")
