"""Weight container (``.sbbw``) and seeded synthetic weights.

The reference loads Keras HDF5 files (``main.py:58-60``, ``main.py:221``).  h5py/Keras are not
available at inference time here, so models travel in a flat container:

    bytes 0..7    magic  b"SBBW0001"
    bytes 8..15   little-endian uint64  header_len
    header        UTF-8 JSON {"model_config": <Keras model_config>, "tensors": [{name, shape, offset}]}
    (zero pad to a 64-byte boundary)
    data          little-endian float32, tensors back to back in header order; offsets in floats

Tensor names and order are exactly Keras' ``weight_names`` (``<layer>/kernel:0`` ...), so an
offline ``.h5`` -> ``.sbbw`` conversion (``tools/h5_to_sbbw.py``) is a straight copy.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Tuple

import numpy as np

from .keras_graph import Graph, parse_model_config, resnet50_unet_config

MAGIC = b"SBBW0001"


def sbbw_bytes(model_config: dict, weights: Dict[str, np.ndarray], graph: Graph = None) -> bytes:
    """The container as one bytes object (what ``sbbseg_model_load`` takes).  ``graph``: the already parsed config, if the caller has it."""
    graph = graph if graph is not None else parse_model_config(model_config)
    tensors, chunks, off = [], [], 0
    for name, shape in graph.weight_specs():
        if name not in weights:
            raise KeyError(f"missing weight {name}")
        w = np.ascontiguousarray(weights[name], dtype="<f4")
        if tuple(w.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {w.shape} != expected {shape}")
        tensors.append({"name": name, "shape": list(shape), "offset": off})
        chunks.append(w.reshape(-1))
        off += w.size
    header = json.dumps({"model_config": model_config, "tensors": tensors}).encode("utf-8")
    pad = (-(16 + len(header))) % 64
    return b"".join([MAGIC, struct.pack("<Q", len(header)), header, b"\0" * pad] + [c.tobytes() for c in chunks])


def save_sbbw(path: str, model_config: dict, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(sbbw_bytes(model_config, weights))


def load_sbbw(path: str) -> Tuple[dict, Dict[str, np.ndarray]]:
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not an SBBW0001 container")
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen).decode("utf-8"))
        f.seek(16 + hlen + ((-(16 + hlen)) % 64))
        data = np.frombuffer(f.read(), dtype="<f4")
    weights = {}
    for t in header["tensors"]:
        n = int(np.prod(t["shape"])) if t["shape"] else 1
        weights[t["name"]] = data[t["offset"]:t["offset"] + n].reshape(t["shape"]).astype(np.float32)
    return header["model_config"], weights


def read_sbbw_config(path: str) -> dict:
    """Only the Keras model_config of a container (the weights stay on disk: the library reads them itself)."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not an SBBW0001 container")
        (hlen,) = struct.unpack("<Q", f.read(8))
        return json.loads(f.read(hlen).decode("utf-8"))["model_config"]


def synthetic_weights(graph: Graph, seed: int = 0) -> Dict[str, np.ndarray]:
    """Seeded stand-in weights: He-normal conv kernels, small biases,
    BN gamma ~ 1 +- 0.1, beta/mean ~ +-0.1, variance ~ 1 +- 0.1.  Deterministic for a given
    (graph, seed) on every platform (PCG64 stream, float64 draw, cast to float32)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shape in graph.weight_specs():
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel:0":
            fan_in = shape[0] * shape[1] * shape[2]
            w = rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif leaf == "bias:0":
            w = rng.uniform(-0.05, 0.05, shape)
        elif leaf == "gamma:0":
            w = rng.uniform(0.9, 1.1, shape)
        elif leaf in ("beta:0", "moving_mean:0"):
            w = rng.uniform(-0.1, 0.1, shape)
        elif leaf == "moving_variance:0":
            w = rng.uniform(0.9, 1.1, shape)
        else:
            raise ValueError(name)
        out[name] = w.astype(np.float32)
    return out


def synthetic_model(n_classes: int = 2, height: int = 448, width: int = 448, seed: int = 0):
    """(model_config, weights) of a synthetic ResNet-50-U-Net -- the stand-in for the three
    external ``.h5`` files of ``main.py:58-60``."""
    cfg = resnet50_unet_config(n_classes, height, width)
    return cfg, synthetic_weights(parse_model_config(cfg), seed)
