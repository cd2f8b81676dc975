"""ORACLE (test infrastructure, not product code) -- fp32 CPU forward pass of a parsed Keras graph.

Stands in for ``model.predict`` of the reference (``main.py:287-288``, ``373-374``), whose
arithmetic lives in keras==2.3.* / tensorflow-gpu==1.15.* (``requirements.txt:5,10``; not vendored,
not installable offline).  Layer semantics restated from their documentation [EXT]:

* ``Conv2D``: NHWC, HWIO kernel, cross-correlation, + bias; ``same`` uses TF's SAME split
  (extra pad goes to bottom/right); ``valid`` none.  -> ``oracle/conv_ref.c``
* ``BatchNormalization`` (inference): ``gamma*(x-mean)/sqrt(var+eps)+beta``, eps from config (1e-3)
* ``ZeroPadding2D``, ``MaxPooling2D`` (valid), ``UpSampling2D`` (nearest: out[y,x]=in[y//f,x//f]),
  ``Concatenate`` (channels), ``Add``, ``Activation`` relu/softmax, ``Lambda`` = one_side_pad crop.

The graph is walked layer by layer, unfused, every tensor fp32 -- deliberately a different
program shape from the fused HIP plan it checks.  ``forward`` accepts the layer list of either reader (the
oracle's own ``keras_config.read_model_config`` or the product's ``keras_graph.parse_model_config``: same
vocabulary); ``OracleModel`` and :func:`forward_config` use the oracle's own.  PARITY UNPINNED vs real Keras/TF (no golden
vectors exist in the reference); cross-checked against torch-CPU in
``tests/test_oracle_forward.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_oracle_lib() -> str:
    so = os.path.join(_HERE, "liboracle_conv.so")
    src = os.path.join(_HERE, "conv_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_conv.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_oracle_lib())
        _LIB.oracle_conv2d_nhwc.restype = ctypes.c_int
        _LIB.oracle_conv2d_nhwc.argtypes = (
            [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] + [ctypes.c_int] * 3 +
            [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p])
        _LIB.oracle_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads() -> int:
    return int(_lib().oracle_num_threads())


def set_num_threads(n: int) -> None:
    _lib().oracle_set_num_threads(int(n))


def conv2d(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], strides, padding: str) -> np.ndarray:
    """x [N,H,W,Cin] f32, w [KH,KW,Cin,Cout] f32 -> [N,Ho,Wo,Cout] f32."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, H, W, Cin = x.shape
    KH, KW, Cin2, Cout = w.shape
    assert Cin == Cin2
    sy, sx = strides
    if padding == "same":
        Ho, Wo = -(-H // sy), -(-W // sx)
        pt = max((Ho - 1) * sy + KH - H, 0) // 2
        pl = max((Wo - 1) * sx + KW - W, 0) // 2
    else:
        Ho, Wo = (H - KH) // sy + 1, (W - KW) // sx + 1
        pt = pl = 0
    y = np.empty((N, Ho, Wo, Cout), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = _lib().oracle_conv2d_nhwc(x.ctypes.data, N, H, W, Cin, w.ctypes.data, KH, KW, Cout,
                                   None if b is None else b.ctypes.data, sy, sx, pt, pl, Ho, Wo, y.ctypes.data)
    if rc != 0:
        raise RuntimeError("oracle_conv2d_nhwc failed")
    return y


def conv2d_transpose(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], strides, padding: str) -> np.ndarray:
    """Keras ``Conv2DTranspose`` [EXT]: x [N,H,W,Cin] f32, w [KH,KW,Cout,Cin] (Keras' transposed-conv kernel layout).
    TF ``conv2d_transpose`` = the gradient of ``conv2d`` w.r.t. its input: every input pixel scatters its
    KH x KW x Cout patch at stride s; 'same' crops the result to H*s x W*s starting max(K-s,0)//2 in (the SAME padding
    of the forward conv it is the gradient of), 'valid' keeps all (H-1)*s + K rows (H*s when K < s).  Plain numpy."""
    x = np.ascontiguousarray(x, np.float32)
    N, H, W, Cin = x.shape
    KH, KW, Cout, Cin2 = w.shape
    assert Cin == Cin2
    sy, sx = strides
    if padding == "same":
        OH, OW, pt, pl = H * sy, W * sx, max(KH - sy, 0) // 2, max(KW - sx, 0) // 2
    else:
        OH, OW, pt, pl = H * sy + max(KH - sy, 0), W * sx + max(KW - sx, 0), 0, 0
    full = np.zeros((N, max((H - 1) * sy + KH, pt + OH), max((W - 1) * sx + KW, pl + OW), Cout), np.float32)
    for ky in range(KH):
        for kx in range(KW):
            full[:, ky:ky + (H - 1) * sy + 1:sy, kx:kx + (W - 1) * sx + 1:sx, :] += x @ np.ascontiguousarray(w[ky, kx].T, np.float32)
    y = full[:, pt:pt + OH, pl:pl + OW, :]
    if bias is not None:
        y = y + np.asarray(bias, np.float32)
    return np.ascontiguousarray(y, np.float32)


def _maxpool(x, pool, strides):
    ph, pw = pool
    sy, sx = strides
    N, H, W, C = x.shape
    Ho, Wo = (H - ph) // sy + 1, (W - pw) // sx + 1
    out = np.full((N, Ho, Wo, C), -np.inf, np.float32)
    for ky in range(ph):
        for kx in range(pw):
            out = np.maximum(out, x[:, ky:ky + (Ho - 1) * sy + 1:sy, kx:kx + (Wo - 1) * sx + 1:sx, :])
    return out


def _softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def forward(graph, weights: Dict[str, np.ndarray], x: np.ndarray, taps=None, conv_hook=None) -> np.ndarray:
    """Run the graph on x [N,H,W,3] (any float dtype; cast to f32 like Keras' predict feed).
    ``taps``: optional dict filled with {layer name: output} for the names it already contains.
    ``conv_hook(node, x, w) -> (x, w)``: lets the error-budget test (tests/test_error_budget.py) perturb the
    operands of chosen Conv2D layers (e.g. round them to fp16) -- never used for parity references."""
    x = np.ascontiguousarray(x, np.float32)
    vals: Dict[str, np.ndarray] = {}
    remaining = {}
    for n in graph.nodes:
        for i in n.inputs:
            remaining[i] = remaining.get(i, 0) + 1
    for n in graph.nodes:
        a = [vals[i] for i in n.inputs]
        if n.op == "input":
            assert tuple(x.shape[1:]) == tuple(n.out_shape), (x.shape, n.out_shape)
            y = x
        elif n.op == "zeropad":
            t, b, l, r = n.attrs["pad"]
            y = np.pad(a[0], ((0, 0), (t, b), (l, r), (0, 0)))
        elif n.op == "conv":
            bias = weights.get(f"{n.name}/bias:0") if n.attrs["use_bias"] else None
            xin, wk = a[0], weights[f"{n.name}/kernel:0"]
            if conv_hook is not None:
                xin, wk = conv_hook(n, xin, wk)
            y = conv2d(xin, wk, bias, n.attrs["strides"], n.attrs["padding"])
            if n.attrs.get("activation", "linear") == "relu":
                y = np.maximum(y, 0)
        elif n.op == "convT":
            bias = weights.get(f"{n.name}/bias:0") if n.attrs["use_bias"] else None
            y = conv2d_transpose(a[0], weights[f"{n.name}/kernel:0"], bias, n.attrs["strides"], n.attrs["padding"])
            if n.attrs.get("activation", "linear") == "relu":
                y = np.maximum(y, 0)
        elif n.op == "bn":
            g = weights[f"{n.name}/gamma:0"] if n.attrs.get("scale", True) else np.float32(1.0)      # Keras: no gamma with scale=False,
            be = weights[f"{n.name}/beta:0"] if n.attrs.get("center", True) else np.float32(0.0)     #        no beta with center=False
            mu = weights[f"{n.name}/moving_mean:0"]; var = weights[f"{n.name}/moving_variance:0"]
            y = (g * (a[0] - mu) / np.sqrt(var + np.float32(n.attrs["eps"])) + be).astype(np.float32)
        elif n.op == "act":
            k = n.attrs["kind"]
            y = np.maximum(a[0], 0) if k == "relu" else _softmax(a[0]) if k == "softmax" else a[0]
        elif n.op == "maxpool":
            y = _maxpool(a[0], n.attrs["pool"], n.attrs["strides"])
        elif n.op == "upsample":
            fy, fx = n.attrs["size"]
            y = np.repeat(np.repeat(a[0], fy, axis=1), fx, axis=2)
        elif n.op == "concat":
            y = np.concatenate(a, axis=3)
        elif n.op == "add":
            y = a[0] + a[1]
        elif n.op == "crop_last":
            y = a[0][:, :-1, :-1, :]
        else:
            raise NotImplementedError(n.op)
        assert tuple(y.shape[1:]) == tuple(n.out_shape), (n.name, y.shape, n.out_shape)
        vals[n.name] = y
        if taps is not None and n.name in taps:
            taps[n.name] = y
        for i in n.inputs:                       # free tensors nobody needs any more
            remaining[i] -= 1
            if remaining[i] == 0 and i != graph.output_name:
                del vals[i]
    return vals[graph.output_name]


def forward_config(model_config, weights: Dict[str, np.ndarray], x: np.ndarray, **kw) -> np.ndarray:
    """forward() on a Keras model_config read by the oracle's own reader (independent of the product's parser)."""
    from .keras_config import read_model_config
    return forward(read_model_config(model_config), weights, x, **kw)


class OracleModel:
    """Duck-typed stand-in for the Keras model (main.py:227-229, 287-288) on the CPU oracle."""

    def __init__(self, model_config, weights):
        from .keras_config import read_model_config          # the oracle's own reader, NOT the product's parser
        self.graph = read_model_config(model_config)
        self.weights = weights
        self.layers = self.graph.nodes

    def predict(self, x):
        return forward(self.graph, self.weights, np.asarray(x))
