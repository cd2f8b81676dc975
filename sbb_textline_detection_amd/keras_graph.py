"""Keras-2.3 ``model_config`` JSON  <->  a small layer-graph IR.

The reference never defines its network: it deserialises three external Keras
HDF5 files (``main.py:58-60``, ``main.py:221``) whose root attribute
``model_config`` is the JSON this module understands.  The files come from the
``resnet50_unet`` definition of qurator-spk/sbb_pixelwise_segmentation
(``README.md:17``).  Two things live here:

* :func:`resnet50_unet_config` -- emits a ``model_config`` dict with the same
  layer classes / wiring / naming conventions Keras 2.3 would serialise for
  that architecture (used to build synthetic models, because no ``.h5`` is
  available offline -- see DESIGN.md "parity unpinned").
* :func:`parse_model_config` -- turns any such JSON into :class:`Graph`, a
  topologically ordered list of :class:`Node` with inferred NHWC output shapes.
  Nothing is executed; ``Lambda`` layers are pattern-matched, never unmarshalled.

The IR is consumed by the oracle interpreter (``oracle/keras_forward.py``,
layer by layer, unfused) and by the planner (``planner.py``, fused for HIP).
"""
from __future__ import annotations

import json
import os
import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

_LAMBDA_WARNED = set()      # layer names whose Lambda lowering has been announced in this process

BN_EPS_DEFAULT = 1e-3  # keras.layers.BatchNormalization default epsilon


# --------------------------------------------------------------------------- IR
@dataclass
class Node:
    name: str
    op: str                       # input|zeropad|conv|convT|bn|act|maxpool|upsample|concat|add|crop_last
    inputs: List[str]
    attrs: Dict = field(default_factory=dict)
    out_shape: Tuple[int, int, int] = (0, 0, 0)   # (H, W, C), batch implied
    class_name: str = ""

    @property
    def output_shape(self):        # Keras-style (None, H, W, C) -- main.py:227-229 reads this
        return (None,) + tuple(self.out_shape)


@dataclass
class Graph:
    nodes: List[Node]
    input_name: str
    output_name: str
    keras_version: str = "2.3.1"

    def by_name(self) -> Dict[str, Node]:
        return {n.name: n for n in self.nodes}

    @property
    def input_shape(self):
        return self.by_name()[self.input_name].out_shape

    @property
    def output_shape(self):
        return self.by_name()[self.output_name].out_shape

    def weight_specs(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """(weight name, shape) in Keras ``weight_names`` order for every layer."""
        byn = self.by_name()
        specs = []
        for n in self.nodes:
            if n.op == "conv":
                cin = byn[n.inputs[0]].out_shape[2]
                kh, kw = n.attrs["kernel"]
                specs.append((f"{n.name}/kernel:0", (kh, kw, cin, n.attrs["filters"])))
                if n.attrs["use_bias"]:
                    specs.append((f"{n.name}/bias:0", (n.attrs["filters"],)))
            elif n.op == "convT":
                cin = byn[n.inputs[0]].out_shape[2]
                kh, kw = n.attrs["kernel"]
                # Keras Conv2DTranspose kernel is (kh, kw, out, in)
                specs.append((f"{n.name}/kernel:0", (kh, kw, n.attrs["filters"], cin)))
                if n.attrs["use_bias"]:
                    specs.append((f"{n.name}/bias:0", (n.attrs["filters"],)))
            elif n.op == "bn":
                c = n.out_shape[2]
                # Keras creates gamma only with scale=True and beta only with center=True (BatchNormalization.build)
                for w in ("gamma", "beta", "moving_mean", "moving_variance"):
                    if (w == "gamma" and not n.attrs.get("scale", True)) or (w == "beta" and not n.attrs.get("center", True)):
                        continue
                    specs.append((f"{n.name}/{w}:0", (c,)))
        return specs


# ------------------------------------------------------------ config builder
class _Builder:
    """Mimics the Keras functional API just enough to serialise ``model_config``."""

    def __init__(self):
        self.layers: List[dict] = []
        self.counters: Dict[str, int] = {}

    def _auto(self, prefix: str) -> str:
        self.counters[prefix] = self.counters.get(prefix, 0) + 1
        return f"{prefix}_{self.counters[prefix]}"

    def _add(self, class_name: str, name: str, config: dict, inbound: Sequence[str]) -> str:
        cfg = {"name": name, "trainable": True}
        cfg.update(config)
        nodes = [[[src, 0, 0, {}] for src in inbound]] if inbound else []
        self.layers.append({"name": name, "class_name": class_name, "config": cfg, "inbound_nodes": nodes})
        return name

    def input(self, h, w, c):
        name = self._auto("input")
        return self._add("InputLayer", name,
                         {"batch_input_shape": [None, h, w, c], "dtype": "float32", "sparse": False}, [])

    def zeropad(self, x, pad):
        return self._add("ZeroPadding2D", self._auto("zero_padding2d"),
                         {"padding": [[pad, pad], [pad, pad]], "data_format": "channels_last"}, [x])

    def conv(self, x, filters, k, strides=1, padding="valid", name=None):
        return self._add("Conv2D", name or self._auto("conv2d"), {
            "filters": filters, "kernel_size": [k, k], "strides": [strides, strides], "padding": padding,
            "data_format": "channels_last", "dilation_rate": [1, 1], "activation": "linear", "use_bias": True,
            "kernel_initializer": {"class_name": "VarianceScaling",
                                   "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}},
            "bias_initializer": {"class_name": "Zeros", "config": {}},
            "kernel_regularizer": {"class_name": "L1L2", "config": {"l1": 0.0, "l2": 1e-6}},
            "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None,
            "bias_constraint": None}, [x])

    def conv_transpose(self, x, filters, k, strides=2, padding="same", name=None):
        return self._add("Conv2DTranspose", name or self._auto("conv2d_transpose"), {
            "filters": filters, "kernel_size": [k, k], "strides": [strides, strides], "padding": padding,
            "output_padding": None, "data_format": "channels_last", "dilation_rate": [1, 1], "activation": "linear",
            "use_bias": True}, [x])

    def bn(self, x, name=None):
        return self._add("BatchNormalization", name or self._auto("batch_normalization"), {
            "axis": 3, "momentum": 0.99, "epsilon": BN_EPS_DEFAULT, "center": True, "scale": True,
            "beta_initializer": {"class_name": "Zeros", "config": {}},
            "gamma_initializer": {"class_name": "Ones", "config": {}},
            "moving_mean_initializer": {"class_name": "Zeros", "config": {}},
            "moving_variance_initializer": {"class_name": "Ones", "config": {}},
            "beta_regularizer": None, "gamma_regularizer": None, "beta_constraint": None,
            "gamma_constraint": None}, [x])

    def act(self, x, kind):
        return self._add("Activation", self._auto("activation"), {"activation": kind}, [x])

    def maxpool(self, x, k, s):
        return self._add("MaxPooling2D", self._auto("max_pooling2d"), {
            "pool_size": [k, k], "padding": "valid", "strides": [s, s], "data_format": "channels_last"}, [x])

    def upsample(self, x, f=2):
        return self._add("UpSampling2D", self._auto("up_sampling2d"), {
            "size": [f, f], "data_format": "channels_last", "interpolation": "nearest"}, [x])

    def concat(self, xs):
        return self._add("Concatenate", self._auto("concatenate"), {"axis": 3}, xs)

    def add(self, xs):
        return self._add("Add", self._auto("add"), {}, xs)

    def lambda_crop_last(self, x):
        # Keras stores marshalled bytecode in "function"; we never evaluate it.  The
        # placeholder keeps the structural shape [code, defaults, closure].
        return self._add("Lambda", self._auto("lambda"), {
            "function": ["<marshalled: lambda x: x[:, :-1, :-1, :]>", None, None],
            "function_type": "lambda", "output_shape": None, "output_shape_type": "raw", "arguments": {}}, [x])


def resnet50_unet_config(n_classes: int, input_height: int = 448, input_width: int = 448) -> dict:
    """``model_config`` of upstream ``resnet50_unet`` (ResNet-50 v1 encoder, stride on the first
    1x1 of each conv_block; decoder = 5 x [UpSampling2D(2) -> concat skip -> ZeroPadding2D(1) ->
    Conv2D 3x3 valid -> BN -> ReLU]; head = Conv2D 1x1 -> BN -> softmax).  SURVEY.md section 8(a-4), 8(d).
    """
    assert input_height % 32 == 0 and input_width % 32 == 0
    b = _Builder()
    img = b.input(input_height, input_width, 3)

    x = b.zeropad(img, 3)
    x = b.conv(x, 64, 7, strides=2, name="conv1")
    f1 = x                                                  # skip taken BEFORE bn_conv1
    x = b.bn(x, name="bn_conv1")
    x = b.act(x, "relu")
    x = b.maxpool(x, 3, 2)

    def conv_block(x, filters, stage, block, strides=2):
        f1_, f2_, f3_ = filters
        cn, bnn = f"res{stage}{block}_branch", f"bn{stage}{block}_branch"
        y = b.conv(x, f1_, 1, strides=strides, name=cn + "2a")
        y = b.bn(y, name=bnn + "2a"); y = b.act(y, "relu")
        y = b.conv(y, f2_, 3, padding="same", name=cn + "2b")
        y = b.bn(y, name=bnn + "2b"); y = b.act(y, "relu")
        y = b.conv(y, f3_, 1, name=cn + "2c")
        y = b.bn(y, name=bnn + "2c")
        s = b.conv(x, f3_, 1, strides=strides, name=cn + "1")
        s = b.bn(s, name=bnn + "1")
        y = b.add([y, s])
        return b.act(y, "relu")

    def identity_block(x, filters, stage, block):
        f1_, f2_, f3_ = filters
        cn, bnn = f"res{stage}{block}_branch", f"bn{stage}{block}_branch"
        y = b.conv(x, f1_, 1, name=cn + "2a")
        y = b.bn(y, name=bnn + "2a"); y = b.act(y, "relu")
        y = b.conv(y, f2_, 3, padding="same", name=cn + "2b")
        y = b.bn(y, name=bnn + "2b"); y = b.act(y, "relu")
        y = b.conv(y, f3_, 1, name=cn + "2c")
        y = b.bn(y, name=bnn + "2c")
        y = b.add([y, x])
        return b.act(y, "relu")

    x = conv_block(x, [64, 64, 256], 2, "a", strides=1)
    for blk in "bc":
        x = identity_block(x, [64, 64, 256], 2, blk)
    f2 = b.lambda_crop_last(b.zeropad(x, 1))                # one_side_pad: 111 -> 113 -> 112

    x = conv_block(x, [128, 128, 512], 3, "a")
    for blk in "bcd":
        x = identity_block(x, [128, 128, 512], 3, blk)
    f3 = x
    x = conv_block(x, [256, 256, 1024], 4, "a")
    for blk in "bcdef":
        x = identity_block(x, [256, 256, 1024], 4, blk)
    f4 = x
    x = conv_block(x, [512, 512, 2048], 5, "a")
    for blk in "bc":
        x = identity_block(x, [512, 512, 2048], 5, blk)
    f5 = x

    o = b.conv(f5, 1024, 1, padding="same")
    o = b.bn(o); o = b.act(o, "relu")
    for skip, filt in ((f4, 512), (f3, 256), (f2, 128), (f1, 64), (img, 32)):
        o = b.upsample(o, 2)
        o = b.concat([o, skip])
        o = b.zeropad(o, 1)
        o = b.conv(o, filt, 3, padding="valid")
        o = b.bn(o); o = b.act(o, "relu")
    o = b.conv(o, n_classes, 1, padding="same")
    o = b.bn(o)
    o = b.act(o, "softmax")

    return {"class_name": "Model",
            "config": {"name": "model_1", "layers": b.layers,
                       "input_layers": [[img, 0, 0]], "output_layers": [[o, 0, 0]]},
            "keras_version": "2.3.1", "backend": "tensorflow"}


def transpose_unet_config(n_classes: int, input_height: int = 64, input_width: int = 64, k: int = 2,
                          padding: str = "same", base: int = 16) -> dict:
    """A small U-Net whose decoder upsamples with ``Conv2DTranspose`` (k x k, stride 2) instead of ``UpSampling2D`` --
    the decoder form BASELINE.json's north_star names ("transposed-conv upsamples"; SURVEY.md 0.5: support both).
    Two 3x3 stride-2 encoder convs, two transposed convs back up, skip concats, 32-channel last conv, 1x1 head."""
    assert input_height % 4 == 0 and input_width % 4 == 0
    b = _Builder()
    img = b.input(input_height, input_width, 3)
    e1 = b.act(b.bn(b.conv(img, base, 3, padding="same")), "relu")                       # H
    e2 = b.act(b.bn(b.conv(e1, 2 * base, 3, strides=2, padding="same")), "relu")         # H/2
    e3 = b.act(b.bn(b.conv(e2, 4 * base, 3, strides=2, padding="same")), "relu")         # H/4
    assert padding == "same" or k == 2, "valid padding with k > stride grows the output (no matching skip size)"
    u = b.act(b.bn(b.conv_transpose(e3, 2 * base, k, 2, padding)), "relu")               # H/2
    o = b.concat([u, e2])
    o = b.act(b.bn(b.conv(o, 2 * base, 3, padding="same")), "relu")
    u = b.act(b.bn(b.conv_transpose(o, base, k, 2, "same")), "relu")                     # H
    o = b.concat([u, e1])
    o = b.act(b.bn(b.conv(o, 32, 3, padding="same")), "relu")
    o = b.act(b.bn(b.conv(o, n_classes, 1, padding="same")), "softmax")
    return {"class_name": "Model",
            "config": {"name": "model_t", "layers": b.layers, "input_layers": [[img, 0, 0]], "output_layers": [[o, 0, 0]]},
            "keras_version": "2.3.1", "backend": "tensorflow"}


# ------------------------------------------------------------------- parser
def _pad4(padding) -> Tuple[int, int, int, int]:
    """Keras ZeroPadding2D ``padding`` -> (top, bottom, left, right)."""
    if isinstance(padding, int):
        return (padding,) * 4
    a, c = padding
    if isinstance(a, int):
        return (a, a, c, c)
    return (a[0], a[1], c[0], c[1])


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    return (int(v[0]), int(v[1]))


def _conv_out(n: int, k: int, s: int, padding: str) -> int:
    if padding == "same":
        return -(-n // s)
    return (n - k) // s + 1


def _same_pad(n: int, k: int, s: int) -> Tuple[int, int]:
    """TF 'SAME' padding (begin, end) for one axis."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def parse_model_config(model_config) -> Graph:
    """Parse a Keras-2.3 functional ``model_config`` (dict or JSON str) into :class:`Graph`."""
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    if model_config.get("class_name") not in ("Model", "Functional"):
        raise ValueError(f"unsupported top-level class {model_config.get('class_name')!r} "
                         "(expected a functional Keras Model)")
    cfg = model_config["config"]
    nodes: List[Node] = []
    shapes: Dict[str, Tuple[int, int, int]] = {}

    for layer in cfg["layers"]:
        cls, name, lc = layer["class_name"], layer["name"], layer["config"]
        inbound = layer.get("inbound_nodes", [])
        if len(inbound) > 1:
            raise ValueError(f"layer {name}: shared layers (multiple inbound nodes) unsupported")
        ins = [t[0] for t in inbound[0]] if inbound else []
        for i in ins:
            if i not in shapes:
                raise ValueError(f"layer {name}: input {i} not yet defined (config not topologically ordered)")
        if lc.get("data_format", "channels_last") != "channels_last":
            raise ValueError(f"layer {name}: only channels_last is supported")
        ish = shapes[ins[0]] if ins else None

        if cls == "InputLayer":
            bis = lc["batch_input_shape"]
            node = Node(name, "input", [], {}, (int(bis[1]), int(bis[2]), int(bis[3])))
        elif cls == "ZeroPadding2D":
            t, bo, l, r = _pad4(lc["padding"])
            node = Node(name, "zeropad", ins, {"pad": (t, bo, l, r)}, (ish[0] + t + bo, ish[1] + l + r, ish[2]))
        elif cls == "Conv2D":
            kh, kw = _pair(lc["kernel_size"]); sy, sx = _pair(lc["strides"])
            if _pair(lc.get("dilation_rate", 1)) != (1, 1):
                raise ValueError(f"layer {name}: dilation unsupported")
            padding = lc["padding"]
            if padding not in ("same", "valid"):
                raise ValueError(f"layer {name}: padding {padding!r} unsupported")
            node = Node(name, "conv", ins, {
                "kernel": (kh, kw), "strides": (sy, sx), "padding": padding, "filters": int(lc["filters"]),
                "use_bias": bool(lc.get("use_bias", True)), "activation": lc.get("activation", "linear")},
                (_conv_out(ish[0], kh, sy, padding), _conv_out(ish[1], kw, sx, padding), int(lc["filters"])))
        elif cls == "Conv2DTranspose":
            kh, kw = _pair(lc["kernel_size"]); sy, sx = _pair(lc["strides"])
            padding = lc["padding"]
            if lc.get("output_padding") is not None or _pair(lc.get("dilation_rate", 1)) != (1, 1) or padding not in ("same", "valid"):
                raise ValueError(f"layer {name}: Conv2DTranspose with output_padding / dilation / padding {padding!r} unsupported")
            oh = ish[0] * sy if padding == "same" else ish[0] * sy + max(kh - sy, 0)
            ow = ish[1] * sx if padding == "same" else ish[1] * sx + max(kw - sx, 0)
            node = Node(name, "convT", ins, {
                "kernel": (kh, kw), "strides": (sy, sx), "padding": padding, "filters": int(lc["filters"]),
                "use_bias": bool(lc.get("use_bias", True)), "activation": lc.get("activation", "linear")},
                (oh, ow, int(lc["filters"])))
        elif cls == "BatchNormalization":
            axis = lc.get("axis", -1)
            axis = axis[0] if isinstance(axis, (list, tuple)) else axis
            if axis not in (3, -1):
                raise ValueError(f"layer {name}: BN axis {axis} unsupported (NHWC only)")
            node = Node(name, "bn", ins, {"eps": float(lc.get("epsilon", BN_EPS_DEFAULT)),
                                          "center": bool(lc.get("center", True)),
                                          "scale": bool(lc.get("scale", True))}, ish)
        elif cls == "Activation":
            kind = lc["activation"]
            if kind not in ("relu", "softmax", "linear"):
                raise ValueError(f"layer {name}: activation {kind!r} unsupported")
            node = Node(name, "act", ins, {"kind": kind}, ish)
        elif cls == "MaxPooling2D":
            ph, pw = _pair(lc["pool_size"]); sy, sx = _pair(lc["strides"] or lc["pool_size"])
            if lc.get("padding", "valid") != "valid":
                raise ValueError(f"layer {name}: only valid max-pooling supported")
            node = Node(name, "maxpool", ins, {"pool": (ph, pw), "strides": (sy, sx)},
                        ((ish[0] - ph) // sy + 1, (ish[1] - pw) // sx + 1, ish[2]))
        elif cls == "UpSampling2D":
            fy, fx = _pair(lc["size"])
            if lc.get("interpolation", "nearest") != "nearest":
                raise ValueError(f"layer {name}: only nearest UpSampling2D supported")
            node = Node(name, "upsample", ins, {"size": (fy, fx)}, (ish[0] * fy, ish[1] * fx, ish[2]))
        elif cls == "Concatenate":
            if lc.get("axis", -1) not in (3, -1):
                raise ValueError(f"layer {name}: only channel concatenation supported")
            hw = {shapes[i][:2] for i in ins}
            if len(hw) != 1:
                raise ValueError(f"layer {name}: concat inputs differ in H,W: {[shapes[i] for i in ins]}")
            node = Node(name, "concat", ins, {}, (ish[0], ish[1], sum(shapes[i][2] for i in ins)))
        elif cls == "Add":
            if len({shapes[i] for i in ins}) != 1:
                raise ValueError(f"layer {name}: add inputs differ in shape")
            node = Node(name, "add", ins, {}, ish)
        elif cls == "Lambda":
            # Upstream's only Lambda is one_side_pad's crop  x[:, :-1, :-1, :]  right after
            # ZeroPadding2D((1,1)).  Match that structure; refuse anything else (never unmarshal).
            # The function body itself is opaque (marshalled CPython bytecode of the training interpreter): a foreign model whose
            # Lambda does something ELSE of the same shape (crops the FIRST row / column, flips, ...) would be lowered wrongly
            # and silently.  What can be checked is checked: position (straight after ZeroPadding2D((1,1))), an anonymous
            # lambda without bound arguments, and a warning that names the layer; SBBSEG_STRICT_LAMBDA=1 refuses instead.
            prev = next(n for n in nodes if n.name == ins[0])
            if prev.op != "zeropad" or prev.attrs["pad"] != (1, 1, 1, 1):
                raise ValueError(f"layer {name}: Lambda not recognised as one_side_pad crop")
            if lc.get("function_type", "lambda") != "lambda" or lc.get("arguments"):
                raise ValueError(f"layer {name}: Lambda with function_type {lc.get('function_type')!r} / arguments {lc.get('arguments')!r} "
                                 f"is not the one_side_pad crop")
            if os.environ.get("SBBSEG_STRICT_LAMBDA", "0") not in ("", "0"):
                raise ValueError(f"layer {name}: Lambda layers are refused (SBBSEG_STRICT_LAMBDA); its bytecode cannot be inspected")
            if name not in _LAMBDA_WARNED:                  # once per process and layer name: a model is parsed several times on its way
                _LAMBDA_WARNED.add(name)                     # to the device (SegModel, the container writer, the plan mirror)
                warnings.warn(f"layer {name}: Lambda after ZeroPadding2D((1,1)) lowered as one_side_pad's crop x[:, :-1, :-1, :] (its bytecode "
                              f"is not inspected; set SBBSEG_STRICT_LAMBDA=1 to refuse)", stacklevel=2)
            node = Node(name, "crop_last", ins, {}, (ish[0] - 1, ish[1] - 1, ish[2]))
        elif cls in ("Dropout", "SpatialDropout2D"):
            node = Node(name, "act", ins, {"kind": "linear"}, ish)     # identity at inference
        else:
            raise ValueError(f"layer {name}: unsupported layer class {cls}")
        node.class_name = cls
        nodes.append(node)
        shapes[name] = node.out_shape

    return Graph(nodes, cfg["input_layers"][0][0], cfg["output_layers"][0][0],
                 model_config.get("keras_version", "?"))
