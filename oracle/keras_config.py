"""ORACLE (test infrastructure, not product code) -- the oracle's OWN reader of a Keras-2.3 ``model_config``.

``oracle/keras_forward.py`` used to walk the graph produced by the product's parser
(``sbb_textline_detection_amd/keras_graph.py``): a wrong padding / concat order / crop attribute in that parser
was then invisible to the product, the oracle and the torch cross-check alike.  This
module is a second, independently written reader: same input (the JSON that ``keras.models.load_model`` parses
out of the ``.h5`` root attribute ``model_config``; reference call site ``main.py:221``), no shared code.  It
returns plain ``Layer`` records that ``keras_forward.forward`` can walk; ``tests/test_keras_config_fixture.py``
compares both readers on a hand-written Keras-2.3 fixture and on the generated ResNet-50-U-Net configs.

Keras-2.3 serialisation facts relied on [EXT, keras/engine/network.py ``get_config``]:
* functional ``Model``: ``config.layers[*] = {name, class_name, config, inbound_nodes}``; an inbound node is a list of
  ``[layer_name, node_index, tensor_index, kwargs]``; layers are listed in creation (= topological) order;
* tuples are serialised as JSON lists (``padding: [[1, 1], [1, 1]]``, ``kernel_size: [3, 3]``); ints stay ints;
* ``Lambda.function`` is ``[base64(marshal(code)), defaults, closure]`` with ``function_type: "lambda"``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import json
from typing import Dict, List, NamedTuple, Tuple


class Layer(NamedTuple):
    name: str
    op: str                      # same vocabulary keras_forward.forward dispatches on
    inputs: Tuple[str, ...]
    attrs: Dict
    out_shape: Tuple[int, int, int]

    @property
    def output_shape(self):      # (None, H, W, C): what main.py:227-229 reads off model.layers[-1]
        return (None,) + tuple(self.out_shape)


class LayerGraph:
    def __init__(self, nodes: List[Layer], input_name: str, output_name: str):
        self.nodes, self.input_name, self.output_name = nodes, input_name, output_name

    def by_name(self):
        return {n.name: n for n in self.nodes}

    @property
    def input_shape(self):
        return self.by_name()[self.input_name].out_shape

    @property
    def output_shape(self):
        return self.by_name()[self.output_name].out_shape


def _two(v) -> Tuple[int, int]:
    return (int(v), int(v)) if not isinstance(v, (list, tuple)) else (int(v[0]), int(v[1]))


def _four_pad(v) -> Tuple[int, int, int, int]:
    """ZeroPadding2D.padding: int | (sym_h, sym_w) | ((top, bottom), (left, right)) -> (t, b, l, r)."""
    if not isinstance(v, (list, tuple)):
        return (int(v),) * 4
    h, w = v
    th, bh = (h, h) if not isinstance(h, (list, tuple)) else h
    lw, rw = (w, w) if not isinstance(w, (list, tuple)) else w
    return int(th), int(bh), int(lw), int(rw)


def _window_out(n: int, k: int, s: int, mode: str) -> int:
    if mode == "same":
        return (n + s - 1) // s
    if mode == "valid":
        return (n - k) // s + 1
    raise ValueError(f"padding mode {mode!r}")


def read_model_config(model_config) -> LayerGraph:
    if isinstance(model_config, (bytes, str)):
        model_config = json.loads(model_config)
    top = model_config["config"]
    shape_of: Dict[str, Tuple[int, int, int]] = {}
    nodes: List[Layer] = []

    def conv(c, i):
        k, s = _two(c["kernel_size"]), _two(c["strides"])
        if _two(c.get("dilation_rate", 1)) != (1, 1):
            raise ValueError("dilated convolution")
        h, w, _ = shape_of[i[0]]
        f = int(c["filters"])
        return "conv", {"kernel": k, "strides": s, "padding": c["padding"], "filters": f, "use_bias": bool(c.get("use_bias", True)),
                        "activation": c.get("activation", "linear")}, (_window_out(h, k[0], s[0], c["padding"]),
                                                                       _window_out(w, k[1], s[1], c["padding"]), f)

    def conv_transpose(c, i):
        k, s = _two(c["kernel_size"]), _two(c["strides"])
        if c.get("output_padding") is not None or _two(c.get("dilation_rate", 1)) != (1, 1) or c["padding"] not in ("same", "valid"):
            raise ValueError("Conv2DTranspose with output_padding / dilation")
        h, w, _ = shape_of[i[0]]
        f = int(c["filters"])
        grow = (0, 0) if c["padding"] == "same" else (max(k[0] - s[0], 0), max(k[1] - s[1], 0))
        return "convT", {"kernel": k, "strides": s, "padding": c["padding"], "filters": f, "use_bias": bool(c.get("use_bias", True)),
                         "activation": c.get("activation", "linear")}, (h * s[0] + grow[0], w * s[1] + grow[1], f)

    def bn(c, i):
        ax = c.get("axis", -1)
        ax = ax[0] if isinstance(ax, (list, tuple)) else ax
        if ax not in (-1, 3):
            raise ValueError("BatchNormalization over a non-channel axis")
        return "bn", {"eps": float(c.get("epsilon", 1e-3)), "center": bool(c.get("center", True)), "scale": bool(c.get("scale", True))}, shape_of[i[0]]

    def zeropad(c, i):
        t, b, l, r = _four_pad(c["padding"])
        h, w, ch = shape_of[i[0]]
        return "zeropad", {"pad": (t, b, l, r)}, (h + t + b, w + l + r, ch)

    def act(c, i):
        if c["activation"] not in ("relu", "softmax", "linear"):
            raise ValueError(f"activation {c['activation']!r}")
        return "act", {"kind": c["activation"]}, shape_of[i[0]]

    def maxpool(c, i):
        p = _two(c["pool_size"])
        s = _two(c["strides"]) if c.get("strides") else p
        if c.get("padding", "valid") != "valid":
            raise ValueError("padded max-pooling")
        h, w, ch = shape_of[i[0]]
        return "maxpool", {"pool": p, "strides": s}, ((h - p[0]) // s[0] + 1, (w - p[1]) // s[1] + 1, ch)

    def upsample(c, i):
        f = _two(c["size"])
        if c.get("interpolation", "nearest") != "nearest":
            raise ValueError("non-nearest UpSampling2D")
        h, w, ch = shape_of[i[0]]
        return "upsample", {"size": f}, (h * f[0], w * f[1], ch)

    def concat(c, i):
        if c.get("axis", -1) not in (-1, 3):
            raise ValueError("concatenation over a non-channel axis")
        hw = {shape_of[n][:2] for n in i}
        if len(hw) != 1:
            raise ValueError("concat inputs differ in size")
        h, w = hw.pop()
        return "concat", {}, (h, w, sum(shape_of[n][2] for n in i))

    def add(c, i):
        if len({shape_of[n] for n in i}) != 1:
            raise ValueError("Add inputs differ in shape")
        return "add", {}, shape_of[i[0]]

    def lam(c, i):
        # upstream's only Lambda (one_side_pad): ZeroPadding2D((1,1)) followed by  x[:, :-1, :-1, :].  The marshalled
        # bytecode is never evaluated; anything that is not directly behind such a padding is refused.
        prev = next(n for n in nodes if n.name == i[0])
        if prev.op != "zeropad" or prev.attrs["pad"] != (1, 1, 1, 1) or c.get("function_type", "lambda") != "lambda":
            raise ValueError("Lambda other than one_side_pad's crop")
        h, w, ch = shape_of[i[0]]
        return "crop_last", {}, (h - 1, w - 1, ch)

    def identity(c, i):
        return "act", {"kind": "linear"}, shape_of[i[0]]

    readers = {"Conv2D": conv, "Conv2DTranspose": conv_transpose, "BatchNormalization": bn, "ZeroPadding2D": zeropad, "Activation": act, "MaxPooling2D": maxpool,
               "UpSampling2D": upsample, "Concatenate": concat, "Add": add, "Lambda": lam, "Dropout": identity,
               "SpatialDropout2D": identity}
    for entry in top["layers"]:
        cls, name, c = entry["class_name"], entry["name"], entry["config"]
        if c.get("data_format", "channels_last") != "channels_last":
            raise ValueError(f"{name}: channels_first")
        inbound = entry.get("inbound_nodes") or []
        if len(inbound) > 1:
            raise ValueError(f"{name}: shared layer")
        ins = tuple(ref[0] for ref in inbound[0]) if inbound else ()
        if cls == "InputLayer":
            shp = c["batch_input_shape"]
            rec = Layer(name, "input", (), {}, (int(shp[1]), int(shp[2]), int(shp[3])))
        elif cls in readers:
            op, attrs, out = readers[cls](c, ins)
            rec = Layer(name, op, ins, attrs, tuple(int(v) for v in out))
        else:
            raise ValueError(f"{name}: layer class {cls} not understood by the oracle")
        nodes.append(rec)
        shape_of[name] = rec.out_shape
    return LayerGraph(nodes, top["input_layers"][0][0], top["output_layers"][0][0])
