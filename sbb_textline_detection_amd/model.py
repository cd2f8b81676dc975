"""Model lifecycle behind the reference's seams (``main.py:216-223``, ``227-229``, ``287-288``).

``start_new_session_and_model(path) -> (model, session)`` mirrors the reference method of the same
name; ``model`` is a :class:`SegModel` with exactly the duck type ``do_prediction`` touches:
``model.layers[-1].output_shape == (None, H, W, C)`` and ``model.predict(x[N,H,W,3]) -> f32
[N,H,W,C]`` softmax probabilities.  Inference runs only through libsbbseg (HIP, gfx950); no
Keras/TF and no CPU fallback.

The reference re-parses its ``.h5`` and rebuilds a TF session for every stage of every page
(``main.py:386, 442, 492``).  Here loaded models are cached per (file, device, precision);
``Session.close()`` only drops a reference and :func:`clear_session` (the ``K.clear_session()``
analogue, ``main.py:2065``) frees device memory.  Set ``SBBSEG_MODEL_CACHE=0`` to free on close.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import _capi
from .keras_graph import Graph, parse_model_config
from .planner import Plan, build_plan
from .weights import load_sbbw, read_sbbw_config, sbbw_bytes

_CACHE: Dict[Tuple, "SegModel"] = {}


def default_max_batch() -> int:
    return int(os.environ.get("SBBSEG_MAX_BATCH", "70"))       # one 3500x2500 page (70 tiles) per chunk


def default_precision() -> str:
    """Arithmetic mode of a model loaded without an explicit ``precision=``.

    ``"f16x3"`` (default) -- the label-exact mode: split-fp16 operands, three MFMAs per product; label maps equal an
    fp32 evaluation of the network except at exact ties (what ``main.py:290`` ``np.argmax`` of the Keras/TF output gives).
    ``"f16"`` -- the fast mode (~3x the throughput): plain fp16 operands; labels may differ where the top-2 softmax margin
    is below ~0.15.  Opt in per call (``precision="f16"``) or process-wide (``SBBSEG_PRECISION=f16``).
    ``"bf16"`` (A/B only) and ``"f32"`` (plain-FMA check mode, slow) exist for tests."""
    return os.environ.get("SBBSEG_PRECISION", "f16x3")


class SegModel:
    """A segmentation net resident on one MI355X, duck-typed like the Keras model the reference uses."""

    def __init__(self, model_config, weights, device: int = 0, max_batch: Optional[int] = None,
                 precision: Optional[str] = None, sbbw_path: Optional[str] = None, planner: Optional[str] = None):
        """The handle is built by the LIBRARY's own graph reader + planner (csrc/loader.cpp) in one C call:
        ``sbbw_path`` -> ``sbbseg_model_load_file`` (the container is read by the library), in-memory ``weights``
        ({Keras weight name: array}) -> serialised to container bytes -> ``sbbseg_model_load``.
        ``planner="python"`` (or ``SBBSEG_PLANNER=python``, or a one-lane handle via ``SBBSEG_LANES=1``) lowers the graph with
        the Python planner (planner.py) instead and uploads the plan step by step: the TEST MIRROR of the native planner --
        the two are compared value for value in tests/test_native_planner.py -- not the product path."""
        precision = precision or default_precision()
        if precision not in _capi.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_capi.PRECISIONS)}, got {precision!r}")
        self.graph: Graph = parse_model_config(model_config)
        self._plan_args = dict(parity_split=os.environ.get("SBBSEG_PARITY_SPLIT", "1") != "0",
                               fuse_head=precision != "f32" and os.environ.get("SBBSEG_FUSE_HEAD", "1") != "0",
                               # the dedicated tail kernels: one-plane 16-bit layout (f16 / bf16) and the split layout (f16x3)
                               fuse_tail=precision in ("f16", "bf16", "f16x3") and os.environ.get("SBBSEG_FUSE_TAIL", "1") != "0",
                               merge_shortcut=os.environ.get("SBBSEG_MERGE_SHORTCUT", "1") != "0")
        self._weights = weights
        self._plan: Optional[Plan] = None
        self.layers = self.graph.nodes                     # main.py:227-229 reads layers[-1].output_shape
        self.device = device
        self.precision = precision
        self.max_batch = int(max_batch or default_max_batch())
        self._sbbw_path = sbbw_path
        prec = _capi.PRECISIONS[precision]
        lanes = int(os.environ.get("SBBSEG_LANES", "2"))
        planner = planner or os.environ.get("SBBSEG_PLANNER", "native")
        if planner not in ("native", "python"):
            raise ValueError(f"planner must be 'native' or 'python', got {planner!r}")
        if sbbw_path is None and weights is None:
            raise ValueError("SegModel needs weights or sbbw_path")
        if lanes != 2 and planner == "native":
            if sbbw_path is not None and weights is None:
                raise ValueError("the native loader builds two-lane handles; use planner='python' for SBBSEG_LANES=1")
            planner = "python"                             # (one-lane handles are an A/B / test configuration)
        self.planner = planner
        if planner == "native":
            a = self._plan_args
            flags = (0 if a["parity_split"] else 1) | (0 if a["merge_shortcut"] else 2) | (0 if a["fuse_head"] or precision == "f32" else 4) | \
                    (0 if a["fuse_tail"] or precision not in ("f16", "bf16", "f16x3") else 8)
            source = sbbw_path if sbbw_path is not None else sbbw_bytes(model_config, weights, self.graph)
            self._ctx: Optional[_capi.Context] = _capi.Context.from_sbbw(source, device, prec, self.max_batch, flags)
        else:
            if weights is None:
                _, self._weights = load_sbbw(sbbw_path)
            self._ctx = _capi.Context(device, prec)
            try:
                self._ctx.set_lanes(lanes)                 # before finalize: 1 skips the second buffer set
                self._ctx.load_plan(self.plan, self.max_batch)
            except Exception:
                self._ctx.close()
                raise
        self._ctx.ids_provider = self.plan_tensor_ids
        self.input_shape = (None,) + tuple(self.graph.input_shape)
        self.output_shape = (None,) + tuple(self.graph.output_shape)

    @property
    def plan(self) -> Plan:
        """The fused op plan as the Python planner builds it (statistics, tests); built on first use for natively loaded models."""
        if self._plan is None:
            if self._weights is None:
                _, self._weights = load_sbbw(self._sbbw_path)
            self._plan = build_plan(self.graph, self._weights, **self._plan_args)
        return self._plan

    # -- seam 2 ------------------------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0):
        """Keras-compatible: float array [N,H,W,3] in [0,1] -> float32 [N,H,W,C] softmax."""
        return self.ctx.predict(np.asarray(x))

    # -- fused fast paths of seam 1 -----------------------------------------------------------
    def segment_page(self, page_u8: np.ndarray, channels: int = 1) -> np.ndarray:
        """uint8 [Hp,Wp,3] -> uint8 [Hp,Wp] label map == do_prediction(True, ...)[:, :, 0]
        (channels=3: uint8 [Hp,Wp,3], the reference's return layout, replicated on the device)."""
        return self.ctx.segment_page(page_u8, channels)

    def segment_whole(self, page_u8: np.ndarray, out_h: int, out_w: int, channels: int = 1) -> np.ndarray:
        """uint8 [Hp,Wp,3] -> uint8 [out_h,out_w] == do_prediction(False, ...)[:, :, 0]."""
        return self.ctx.segment_whole(page_u8, out_h, out_w, channels)

    @property
    def ctx(self) -> _capi.Context:
        if self._ctx is None or self._ctx.h is None:
            raise RuntimeError("model has been released (session closed)")
        return self._ctx

    def plan_tensor_ids(self):
        """Library tensor id of every plan tensor (debug reads of intermediate activations).  The native planner creates its
        tensors in the Python plan's order (ids are sequential; the two input forms are created once), so for a natively
        loaded handle the ids are re-derived from the mirror plan."""
        if self._ctx.tensor_ids:
            return self._ctx.tensor_ids
        ids, nxt, forms = [], 0, {}
        for t in self.plan.tensors:
            if t.kind == "unused":
                ids.append(-1)
            elif t.kind in ("input_c8", "input_pairs"):
                if t.kind not in forms:
                    forms[t.kind] = nxt
                    nxt += 1
                ids.append(forms[t.kind])
            else:
                ids.append(nxt)
                nxt += 1
        self._ctx.tensor_ids = ids
        return ids

    def release(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


class Session:
    """Stand-in for the ``tf.InteractiveSession`` the reference closes after each stage (main.py:428)."""

    def __init__(self, model: SegModel, cached: bool):
        self._model, self._cached, self.closed = model, cached, False

    def close(self):
        if self.closed:
            return                       # double close is harmless
        self.closed = True
        if not self._cached and self._model is not None:
            self._model.release()
        self._model = None


def resolve_model_path(path: str) -> str:
    """The reference is handed ``<dir>/model_*.h5`` (main.py:58-60).  HDF5 cannot be read at inference
    time here; the offline converter (tools/h5_to_sbbw.py) writes ``<same name>.sbbw`` next to it."""
    if path.endswith(".sbbw") and os.path.exists(path):
        return path
    alt = os.path.splitext(path)[0] + ".sbbw"
    if os.path.exists(alt):
        return alt
    if os.path.exists(path):
        raise RuntimeError(f"{path}: Keras HDF5 must be converted once with tools/h5_to_sbbw.py "
                           f"(expected {alt})")
    raise FileNotFoundError(path)


def load_model(path: str, compile: bool = False, device: int = 0, max_batch: Optional[int] = None,
               precision: Optional[str] = None) -> SegModel:
    """``keras.models.load_model(path, compile=False)`` replacement (main.py:221).  ``precision``: see
    :func:`default_precision` (label-exact ``"f16x3"`` unless the caller opts into the fast ``"f16"``)."""
    precision = precision or default_precision()
    real = resolve_model_path(path)
    use_cache = os.environ.get("SBBSEG_MODEL_CACHE", "1") != "0"
    key = (os.path.realpath(real), os.path.getmtime(real), device, precision, int(max_batch or default_max_batch()))
    if use_cache and key in _CACHE and _CACHE[key]._ctx is not None:
        return _CACHE[key]
    if os.environ.get("SBBSEG_NATIVE_LOADER", "1") != "0" and os.environ.get("SBBSEG_LANES", "2") == "2":
        # one C call: the library reads the container and lowers the graph itself (sbbseg_model_load_file)
        model = SegModel(read_sbbw_config(real), None, device=device, max_batch=max_batch, precision=precision, sbbw_path=real)
    else:
        # SBBSEG_NATIVE_LOADER=0 (or a one-lane handle): the Python planner lowers the graph and uploads the plan step by step --
        # the test mirror of csrc/loader.cpp, selectable for A/B; the weights are held once (no container bytes, no C-side copy)
        cfg, weights = load_sbbw(real)
        model = SegModel(cfg, weights, device=device, max_batch=max_batch, precision=precision, planner="python")
    if use_cache:
        _CACHE[key] = model
    model._from_cache = use_cache
    return model


def start_new_session_and_model(model_dir: str, **kw):
    """Same name / argument / return as ``textline_detector.start_new_session_and_model`` (main.py:216-223)."""
    model = load_model(model_dir, compile=False, **kw)
    return model, Session(model, getattr(model, "_from_cache", False))


def clear_session():
    """``K.clear_session()`` analogue (main.py:2065...): free every cached model's device memory."""
    for m in list(_CACHE.values()):
        m.release()
    _CACHE.clear()
