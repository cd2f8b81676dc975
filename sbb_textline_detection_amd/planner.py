"""Lower a parsed Keras graph (``keras_graph.Graph``) into the fused op list libsbbseg executes.

What the reference gets from ``keras.models.load_model`` + ``model.predict`` (``main.py:221,
287-288``) -- ~210 Keras layers run one by one -- becomes ~60 kernels here:

* ``Conv2D -> BatchNormalization -> [Add] -> relu``        one conv, BN/bias folded to fp32
  scale/shift, residual + ReLU in the epilogue
* ``UpSampling2D -> Concatenate -> ZeroPadding2D -> Conv2D``  address arithmetic of that conv's gather
* ``ZeroPadding2D -> Lambda(x[:, :-1, :-1, :])`` (one_side_pad)  a (1,1) placement offset
* the stem ``ZeroPadding2D(3) -> Conv2D 7x7 s2`` on the 3-channel image  a 7x4 stride-(2,1) conv
  over the PAIRS input form (two horizontal neighbours x 4 channels per 16-byte granule)
* final ``Conv2D 1x1 -> BN -> softmax`` (+ the caller's ``np.argmax``, ``main.py:290``)  the head op

The result is plain data (``Plan``); ``_capi.build_context`` feeds it to the C ABI, and
``tests/plan_interp.py`` can interpret it with numpy to check the lowering against the oracle
without a GPU.  Nothing here runs arithmetic on activations.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .keras_graph import Graph, Node

INPUT_C8, INPUT_PAIRS = 0, 1


@dataclass
class Seg:
    tensor: int          # plan tensor id
    channels: int
    shift: int = 0       # nearest x2 upsampling fused into the gather
    off_y: int = 0
    off_x: int = 0
    # conv geometry of this source (filled when the segment becomes a conv operand)
    kh: int = 0
    kw: int = 0
    stride_y: int = 1
    stride_x: int = 1
    pad_top: int = 0
    pad_left: int = 0
    w: Optional[np.ndarray] = None      # float32 [kh][kw][channels][cout]


@dataclass
class TensorSpec:
    H: int
    W: int
    C: int
    kind: str = "act"    # act | input_c8 | input_pairs
    pad: int = 0
    name: str = ""


@dataclass
class ConvStep:
    name: str
    srcs: List[Seg]               # 1 or 2 sources, each with its own taps / stride / padding / weights
    cout: int
    out_h: int                    # output grid of this step ...
    out_w: int
    scale: np.ndarray             # float32 [cout]
    shift: np.ndarray
    out_stride: Tuple[int, int] = (1, 1)   # ... and its placement in the output tensor:
    out_off: Tuple[int, int] = (0, 0)      #     tensor[y*stride+off] = result[y]
    out: int = -1
    relu: bool = False
    residual: int = -1
    raw_out: int = -1
    raw_scale: Optional[np.ndarray] = None
    raw_shift: Optional[np.ndarray] = None
    head: Optional["HeadStep"] = None      # fused network head (1x1 conv + BN + softmax + argmax)
    algorithmic_macs: float = 0.0          # MACs per patch in the reference's formulation
    kind: str = "conv"


@dataclass
class PoolStep:
    name: str
    src: int
    dst: int
    k: int
    stride: int
    pre_scale: Optional[np.ndarray] = None    # per-channel affine + ReLU applied before the max
    pre_shift: Optional[np.ndarray] = None
    pre_relu: bool = False
    kind: str = "maxpool"


@dataclass
class HeadStep:
    name: str
    src: int
    cin: int
    classes: int
    w: np.ndarray                 # float32 [cin][classes]
    scale: np.ndarray
    shift: np.ndarray
    kind: str = "head"


@dataclass
class TailStep:
    """Fused network tail: ReLU(BN(conv3x3 over [up2(src0: 64 ch), image 3 ch])) -> head, one kernel."""
    name: str
    src0: int                     # 64-channel tensor at half resolution
    img: int                      # C8 input form
    w_src0: np.ndarray            # float32 [3][3][64][32] original taps
    w_img: np.ndarray             # float32 [3][3][3][32]
    scale: np.ndarray
    shift: np.ndarray
    head: HeadStep
    out_h: int
    out_w: int
    algorithmic_macs: float
    kind: str = "tail"


@dataclass
class Plan:
    in_h: int
    in_w: int
    classes: int
    tensors: List[TensorSpec] = field(default_factory=list)
    steps: List[object] = field(default_factory=list)
    layer_tensor: Dict[str, int] = field(default_factory=dict)   # Keras layer name -> materialised tensor

    def macs_per_patch(self) -> int:
        """MACs of the reference's formulation (what the roofline counts as algorithmic work)."""
        total = 0
        for s in self.steps:
            if s.kind == "conv":
                total += s.algorithmic_macs
                if s.head is not None:
                    total += s.out_h * s.out_w * s.head.cin * s.head.classes
            elif s.kind == "tail":
                total += s.algorithmic_macs + s.out_h * s.out_w * s.head.cin * s.head.classes
            elif s.kind == "head":
                t = self.tensors[s.src]
                total += t.H * t.W * s.cin * s.classes
        return int(round(total))

    def executed_macs_per_patch(self) -> int:
        """MACs the kernels actually issue (parity-split decoder convs pre-sum coincident taps)."""
        total = 0
        for s in self.steps:
            if s.kind == "conv":
                total += s.out_h * s.out_w * s.cout * sum(g.kh * g.kw * g.channels for g in s.srcs)
                if s.head is not None:
                    total += s.out_h * s.out_w * s.head.cin * s.head.classes
            elif s.kind == "tail":
                total += s.out_h * s.out_w * (32 * (4 * 64 + 9 * 3) + s.head.cin * s.head.classes)
            elif s.kind == "head":
                t = self.tensors[s.src]
                total += t.H * t.W * s.cin * s.classes
        return int(total)


class _Pending:
    """A conv whose epilogue is still being fused."""

    def __init__(self, node: Node, srcs, kh, kw, sy, sx, pt, pl, cout, w, bias, out_hw, logical_taps):
        self.node = node
        self.srcs, self.kh, self.kw, self.sy, self.sx, self.pt, self.pl = srcs, kh, kw, sy, sx, pt, pl
        self.cout, self.w = cout, w
        self.scale = np.ones(cout, np.float64)
        self.shift = np.zeros(cout, np.float64) if bias is None else bias.astype(np.float64)
        self.relu = False
        self.residual = -1
        self.raw_needed = False
        self.raw_scale = self.scale.copy()
        self.raw_shift = self.shift.copy()
        self.out_hw = out_hw
        self.logical_macs_per_out = logical_taps
        self.emitted_out = -1
        self.emitted_raw = -1
        self.stage = "conv"          # conv -> bn -> (add) -> relu
        self.multi = None            # merged convs: explicit per-source Segs (geometry + weights)
        self.convT = None            # Conv2DTranspose: (kh, kw, pad_top, pad_left) of the transposed conv, stride 2
        self.name = node.name


class _View:
    def __init__(self, H, W, segs=None, pad=(0, 0, 0, 0), pending=None, raw_of=None):
        self.H, self.W = H, W
        self.segs: List[Seg] = segs or []
        self.pad = pad                   # pending zero padding (t, b, l, r); H, W include it
        self.pending: Optional[_Pending] = pending
        self.raw_of: Optional[_Pending] = raw_of   # the un-normalised output of a pending conv


class PlanError(NotImplementedError):
    pass


# taps of a 3x3 window (index ky = dy+1) that land on source row a+t (+py-1) of a nearest-x2-upsampled
# tensor, for output-row parity py:  ((2a+py) + dy) >> 1
_PARITY_TAPS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}


def build_plan(graph: Graph, weights: Dict[str, np.ndarray], parity_split: bool = True,
               fuse_head: bool = True, fuse_tail: bool = True, merge_shortcut: bool = True) -> Plan:
    byn = graph.by_name()
    consumers: Dict[str, int] = {}
    for n in graph.nodes:
        for i in n.inputs:
            consumers[i] = consumers.get(i, 0) + 1

    in_h, in_w, in_c = graph.input_shape
    if in_c != 3:
        raise PlanError("network input must have 3 channels")
    plan = Plan(in_h, in_w, 0)
    views: Dict[str, _View] = {}
    input_forms: Dict[Tuple[int, int], int] = {}

    def new_tensor(H, W, C, name="", kind="act", pad=0):
        plan.tensors.append(TensorSpec(H, W, C, kind, pad, name))
        return len(plan.tensors) - 1

    def input_form(form, pad):
        key = (form, pad)
        if key not in input_forms:
            if form == INPUT_C8:
                input_forms[key] = new_tensor(in_h, in_w, 8, "input_c8", "input_c8", 0)
            else:
                if any(k[0] == INPUT_PAIRS for k in input_forms):
                    raise PlanError("only one PAIRS input form (one padding) is supported")
                input_forms[key] = new_tensor(in_h + 2 * pad, (in_w + 2 * pad + 1) // 2, 8, "input_pairs", "input_pairs", pad)
        return input_forms[key]

    def emit(p: _Pending):
        if p.emitted_out >= 0 or p.emitted_raw >= 0:
            return
        oh, ow = p.out_hw
        p.emitted_out = new_tensor(oh, ow, p.cout, p.name)
        if p.raw_needed:
            p.emitted_raw = new_tensor(oh, ow, p.cout, p.node.name + ":raw")
        cbase, srcs = 0, []
        if p.multi is not None:
            srcs = list(p.multi)                       # sources with their own geometry and (pre-scaled) weights
        else:
            for g in p.srcs:
                srcs.append(Seg(g.tensor, g.channels, g.shift, g.off_y, g.off_x, p.kh, p.kw, p.sy, p.sx, p.pt, p.pl,
                                np.ascontiguousarray(p.w[:, :, cbase:cbase + g.channels, :], np.float32)))
                cbase += g.channels
        common = dict(cout=p.cout, scale=p.scale.astype(np.float32), shift=p.shift.astype(np.float32),
                      out=p.emitted_out, relu=p.relu, residual=p.residual, raw_out=p.emitted_raw,
                      raw_scale=p.raw_scale.astype(np.float32) if p.raw_needed else None,
                      raw_shift=p.raw_shift.astype(np.float32) if p.raw_needed else None)
        if p.convT is not None:
            # Conv2DTranspose, stride 2:  out[2i + ky - pt][2j + kx - pl] += x[i][j] . w[ky][kx]   (gradient-of-conv form).
            # Output row y = 2a + py collects the taps ky with (py + pt - ky) even, from input row a + d, d = (py + pt - ky) / 2:
            # each output-parity class is an ordinary stride-1 conv over the SOURCE resolution with a sub-kernel of the taps
            # that land on it (k = 2: four 1x1 convs; k = 3: 2x2 / 2x1 / 1x2 / 1x1) -- same MACs as the reference's scatter form.
            kh, kw, pt, pl = p.convT
            ih, iw = p.in_hw
            macs_t = float(ih * iw * kh * kw * sum(g.channels for g in srcs) * p.cout)
            n_cls = 0
            for py in (0, 1):
                dys = sorted({(py + pt - ky) // 2 for ky in range(kh) if (py + pt - ky) % 2 == 0})
                for px in (0, 1):
                    dxs = sorted({(px + pl - kx) // 2 for kx in range(kw) if (px + pl - kx) % 2 == 0})
                    ch, cw = (oh - py + 1) // 2, (ow - px + 1) // 2
                    if ch <= 0 or cw <= 0:
                        continue
                    if not dys or not dxs:
                        raise PlanError(f"{p.name}: Conv2DTranspose parity class ({py},{px}) receives no taps (kernel smaller than stride)")
                    kh2, kw2 = dys[-1] - dys[0] + 1, dxs[-1] - dxs[0] + 1
                    psrcs = []
                    for g in srcs:
                        w2 = np.zeros((kh2, kw2, g.channels, p.cout), np.float32)
                        for ty in range(kh2):
                            ky = py + pt - 2 * (dys[0] + ty)
                            for tx in range(kw2):
                                kx = px + pl - 2 * (dxs[0] + tx)
                                if 0 <= ky < kh and 0 <= kx < kw:
                                    w2[ty, tx] = g.w[ky, kx]
                        psrcs.append(Seg(g.tensor, g.channels, 0, g.off_y, g.off_x, kh2, kw2, 1, 1, -dys[0], -dxs[0], w2))
                    plan.steps.append(ConvStep(f"{p.node.name}:t{py}{px}", psrcs, out_h=ch, out_w=cw, out_stride=(2, 2),
                                               out_off=(py, px), algorithmic_macs=0.0, **common))
                    n_cls += 1
            for st in plan.steps[-n_cls:]:
                st.algorithmic_macs = macs_t / n_cls
            return
        macs = float(oh * ow * p.cout * p.logical_macs_per_out)
        origin = dict(srcs=srcs, geom=(p.kh, p.kw, p.sy, p.sx, p.pt, p.pl), out_hw=(oh, ow), macs=macs, name=p.node.name)
        splittable = (parity_split and p.multi is None and srcs[0].shift == 1 and (p.kh, p.kw, p.sy, p.sx, p.pt, p.pl) == (3, 3, 1, 1, 1, 1)
                      and oh % 2 == 0 and ow % 2 == 0 and p.residual < 0 and not p.raw_needed
                      and all(not (g.shift and (g.off_y or g.off_x)) for g in srcs))
        if not splittable:
            plan.steps.append(ConvStep(p.name, srcs, out_h=oh, out_w=ow, algorithmic_macs=macs, **common))
            plan.steps[-1].origin = origin
            return
        # 3x3 conv over nearest-x2-upsampled sources == four output-parity classes; in each, the taps
        # that read the same stored pixel are pre-summed: a 2x2 conv at the source's own resolution
        # (2.25x fewer MACs on that source).  Non-upsampled sources become 3x3 stride-2 convs.
        for py in (0, 1):
            for px in (0, 1):
                psrcs = []
                for g in srcs:
                    if g.shift == 1:
                        w2 = np.zeros((2, 2, g.channels, p.cout), np.float64)
                        for ty in (0, 1):
                            for tx in (0, 1):
                                for ky in _PARITY_TAPS[(py, ty)]:
                                    for kx in _PARITY_TAPS[(px, tx)]:
                                        w2[ty, tx] += g.w[ky, kx]
                        psrcs.append(Seg(g.tensor, g.channels, 0, 0, 0, 2, 2, 1, 1, 1 - py, 1 - px, w2.astype(np.float32)))
                    else:
                        psrcs.append(Seg(g.tensor, g.channels, 0, g.off_y, g.off_x, 3, 3, 2, 2, 1 - py, 1 - px, g.w))
                plan.steps.append(ConvStep(f"{p.node.name}:p{py}{px}", psrcs, out_h=oh // 2, out_w=ow // 2,
                                           out_stride=(2, 2), out_off=(py, px), algorithmic_macs=macs / 4, **common))
                plan.steps[-1].origin = origin

    def materialize(name: str) -> _View:
        """Make the value of Keras layer `name` a plain stored tensor (emit pending work)."""
        v = views[name]
        if v.pending is not None:
            emit(v.pending)
            v.segs = [Seg(v.pending.emitted_out, v.pending.cout)]
            v.pending = None
        elif v.raw_of is not None:
            p = v.raw_of
            if p.emitted_out >= 0 and p.emitted_raw < 0:
                raise PlanError(f"{name}: raw conv output requested after the conv was emitted")
            p.raw_needed = True
            emit(p)
            v.segs = [Seg(p.emitted_raw, p.cout)]
            v.raw_of = None
        if v.segs and len(v.segs) == 1 and v.pad == (0, 0, 0, 0) and not v.segs[0].shift \
                and not v.segs[0].off_y and not v.segs[0].off_x and v.segs[0].tensor >= 0:
            plan.layer_tensor[name] = v.segs[0].tensor
        return v

    def plain_tensor(name: str) -> int:
        v = materialize(name)
        s = v.segs
        if len(s) != 1 or s[0].shift or s[0].off_y or s[0].off_x or v.pad != (0, 0, 0, 0):
            raise PlanError(f"{name}: needs a plain stored tensor here")
        return s[0].tensor

    def gatherable(name: str) -> _View:
        """Value usable as a conv source: stored segments with shift/offset/pad folded."""
        v = materialize(name)
        t, b, l, r = v.pad
        segs = [Seg(s.tensor, s.channels, s.shift, s.off_y + t, s.off_x + l) for s in v.segs]
        return _View(v.H, v.W, segs)

    for n in graph.nodes:
        if n.op == "input":
            views[n.name] = _View(in_h, in_w, [Seg(-1, 3)])        # tensor -1 = the network input
        elif n.op == "zeropad":
            src = materialize(n.inputs[0])
            t, b, l, r = n.attrs["pad"]
            pt, pb, pl, pr = src.pad
            views[n.name] = _View(src.H + t + b, src.W + l + r, src.segs, (pt + t, pb + b, pl + l, pr + r))
        elif n.op == "crop_last":
            src = views[n.inputs[0]]
            t, b, l, r = src.pad
            if b < 1 or r < 1:
                raise PlanError(f"{n.name}: crop of real data unsupported")
            views[n.name] = _View(src.H - 1, src.W - 1, src.segs, (t, b - 1, l, r - 1))
        elif n.op == "upsample":
            if n.attrs["size"] != (2, 2):
                raise PlanError(f"{n.name}: only x2 upsampling supported")
            src = gatherable(n.inputs[0])
            if any(s.shift or s.off_y or s.off_x for s in src.segs):
                raise PlanError(f"{n.name}: upsampling of an already transformed view")
            views[n.name] = _View(src.H * 2, src.W * 2, [Seg(s.tensor, s.channels, 1) for s in src.segs])
        elif n.op == "concat":
            parts = [gatherable(i) for i in n.inputs]
            if len({(p.H, p.W) for p in parts}) != 1:
                raise PlanError(f"{n.name}: concat inputs differ in size")
            views[n.name] = _View(parts[0].H, parts[0].W, [s for p in parts for s in p.segs])
        elif n.op == "conv":
            src = gatherable(n.inputs[0])
            kh, kw = n.attrs["kernel"]
            sy, sx = n.attrs["strides"]
            oh, ow, cout = n.out_shape
            if n.attrs["padding"] == "same":
                pt = max((oh - 1) * sy + kh - src.H, 0) // 2
                pl = max((ow - 1) * sx + kw - src.W, 0) // 2
            else:
                pt = pl = 0
            w = weights[f"{n.name}/kernel:0"].astype(np.float32)
            bias = weights[f"{n.name}/bias:0"] if n.attrs["use_bias"] else None
            segs = src.segs
            if len(segs) > 2:
                raise PlanError(f"{n.name}: more than two concatenated sources")
            # a placement offset shared by every source is just conv padding
            my, mx = min(s.off_y for s in segs), min(s.off_x for s in segs)
            segs = [Seg(s.tensor, s.channels, s.shift, s.off_y - my, s.off_x - mx) for s in segs]
            pt, pl = pt + my, pl + mx
            if any(s.shift and (s.off_y or s.off_x) for s in segs):
                raise PlanError(f"{n.name}: upsampled source with a placement offset unsupported")
            taps = kh * kw * sum(s.channels for s in segs)
            if any(s.tensor == -1 for s in segs):
                # a conv that reads the image itself
                segs = list(segs)
                for k, s in enumerate(segs):
                    if s.tensor != -1:
                        continue
                    if s.shift:
                        raise PlanError(f"{n.name}: upsampled network input unsupported")
                    if sx == 2 and len(segs) == 1:
                        # stem: fold the horizontal stride into 2-pixel granules of the PAIRS form
                        pad = s.off_y + pt
                        if pad != s.off_x + pl:
                            raise PlanError(f"{n.name}: asymmetric stem padding unsupported")
                        t_id = input_form(INPUT_PAIRS, pad)
                        kw2 = (kw + 1) // 2
                        w2 = np.zeros((kh, kw2, 8, cout), np.float32)
                        for dx in range(kw):
                            w2[:, dx // 2, (dx & 1) * 4:(dx & 1) * 4 + 3, :] = w[:, dx, :, :]
                        segs[k] = Seg(t_id, 8, 0, 0, 0)
                        w, kw, sx, pt, pl = w2, kw2, 1, 0, 0
                    elif sx == 1:
                        segs[k] = Seg(input_form(INPUT_C8, 0), 3, 0, s.off_y, s.off_x)
                    else:
                        raise PlanError(f"{n.name}: unsupported conv on the network input")
            p = _Pending(n, segs, kh, kw, sy, sx, pt, pl, cout, w, bias, (oh, ow), taps)
            if n.attrs.get("activation", "linear") == "relu":
                p.relu, p.stage = True, "relu"
            elif n.attrs.get("activation", "linear") != "linear":
                raise PlanError(f"{n.name}: inline activation {n.attrs['activation']} unsupported")
            views[n.name] = _View(oh, ow, pending=p)
        elif n.op == "bn":
            src = views[n.inputs[0]]
            p = src.pending
            if p is None or p.stage != "conv":
                raise PlanError(f"{n.name}: BatchNormalization must follow a Conv2D directly")
            if consumers.get(n.inputs[0], 0) > 1:
                # somebody else wants the un-normalised conv output (the f1 skip): keep both
                p.raw_needed = True
                views[n.inputs[0]] = _View(src.H, src.W, raw_of=p)
            g = weights[f"{n.name}/gamma:0"].astype(np.float64) if n.attrs["scale"] else 1.0
            be = weights[f"{n.name}/beta:0"].astype(np.float64) if n.attrs["center"] else 0.0
            mu = weights[f"{n.name}/moving_mean:0"].astype(np.float64)
            var = weights[f"{n.name}/moving_variance:0"].astype(np.float64)
            a = g / np.sqrt(var + n.attrs["eps"])
            p.scale, p.shift = p.scale * a, (p.shift - mu) * a + be
            p.stage = "bn"
            views[n.name] = _View(src.H, src.W, pending=p)
        elif n.op == "add":
            if len(n.inputs) != 2:
                raise PlanError(f"{n.name}: Add of {len(n.inputs)} inputs unsupported")
            va, vb = views[n.inputs[0]], views[n.inputs[1]]
            cand = [k for k, v in ((0, va), (1, vb))
                    if v.pending is not None and v.pending.stage in ("conv", "bn") and not v.pending.relu
                    and v.pending.residual < 0 and consumers.get(n.inputs[k], 0) == 1]
            if not cand:
                raise PlanError(f"{n.name}: Add needs one input that is a conv(+BN) with a single consumer")
            pa, pb = va.pending, vb.pending
            if (merge_shortcut and len(cand) == 2 and pa.multi is None and pb.multi is None and pa.convT is None and pb.convT is None
                    and len(pa.srcs) == 1
                    and len(pb.srcs) == 1 and not pa.raw_needed and not pb.raw_needed and pa.cout == pb.cout
                    and pa.out_hw == pb.out_hw and not pa.srcs[0].shift and not pb.srcs[0].shift
                    and pa.srcs[0].tensor >= 0 and pb.srcs[0].tensor >= 0):
                # projection-shortcut block: BN_a(conv_a(x)) + BN_b(conv_b(y)) is ONE conv over two sources
                # with the BN scales folded into the weight rows (exact algebra):
                #   sum_k (s_a W_a)[c,k] x[k] + sum_k (s_b W_b)[c,k] y[k] + (shift_a + shift_b)
                # -> the shortcut tensor is never written or re-read as a residual, one launch instead of two
                segs = []
                for q in (pa, pb):
                    g = q.srcs[0]
                    wq = (q.w.astype(np.float64) * q.scale[None, None, None, :]).astype(np.float32)
                    segs.append(Seg(g.tensor, g.channels, 0, g.off_y, g.off_x, q.kh, q.kw, q.sy, q.sx, q.pt, q.pl,
                                    np.ascontiguousarray(wq)))
                pa.multi = segs
                pa.srcs = [Seg(g.tensor, g.channels, 0, g.off_y, g.off_x) for g in segs]
                pa.shift = pa.shift + pb.shift
                pa.scale = np.ones(pa.cout, np.float64)
                pa.logical_macs_per_out += pb.logical_macs_per_out
                pa.name = f"{pa.node.name}+{pb.node.name}"
                pa.stage = "add"
                views[n.name] = _View(va.H, va.W, pending=pa)
                continue
            k = cand[0]
            other = plain_tensor(n.inputs[1 - k])
            p = views[n.inputs[k]].pending
            ot = plan.tensors[other]
            if (ot.H, ot.W, ot.C) != (p.out_hw[0], p.out_hw[1], p.cout):
                raise PlanError(f"{n.name}: residual shape mismatch")
            p.residual, p.stage = other, "add"
            views[n.name] = _View(va.H, va.W, pending=p)
        elif n.op == "act":
            kind = n.attrs["kind"]
            src = views[n.inputs[0]]
            if kind == "linear":
                views[n.name] = src
            elif kind == "relu":
                p = src.pending
                if p is None or p.relu or consumers.get(n.inputs[0], 0) != 1:
                    raise PlanError(f"{n.name}: relu must follow conv/BN/Add with a single consumer")
                p.relu, p.stage = True, "relu"
                views[n.name] = _View(src.H, src.W, pending=p)
            elif kind == "softmax":
                p = src.pending
                if (n.name != graph.output_name or p is None or p.relu or p.residual >= 0 or (p.kh, p.kw) != (1, 1)
                        or len(p.srcs) != 1 or p.srcs[0].shift or p.srcs[0].off_y or p.srcs[0].off_x
                        or (p.sy, p.sx) != (1, 1)):
                    raise PlanError(f"{n.name}: softmax is only supported as the final 1x1-conv head")
                t_id = p.srcs[0].tensor
                ts = plan.tensors[t_id]
                if ts.C != p.srcs[0].channels or ts.C > 64 or p.cout > 8 or (ts.H, ts.W) != (in_h, in_w):
                    raise PlanError(f"{n.name}: head needs <=64 input channels, <=8 classes, input resolution")
                head = HeadStep(p.node.name, t_id, ts.C, p.cout,
                                np.ascontiguousarray(p.w.reshape(ts.C, p.cout), np.float32),
                                p.scale.astype(np.float32), p.shift.astype(np.float32))
                producers = [st for st in plan.steps if st.kind == "conv" and st.out == t_id]
                src_layer = p.node.inputs[0]
                fusable = (fuse_head and producers and ts.C == 32 and p.cout <= 4 and consumers.get(src_layer, 0) == 1
                           and all(st.raw_out < 0 and st.residual < 0 for st in producers))
                org = getattr(producers[0], "origin", None) if producers else None
                if (fusable and fuse_tail and org is not None and org["geom"] == (3, 3, 1, 1, 1, 1) and len(org["srcs"]) == 2
                        and org["srcs"][0].shift == 1 and org["srcs"][0].channels == 64
                        and plan.tensors[org["srcs"][0].tensor].C == 64
                        and plan.tensors[org["srcs"][1].tensor].kind == "input_c8" and org["srcs"][1].channels == 3
                        and not org["srcs"][1].off_y and not org["srcs"][1].off_x and producers[0].relu
                        and org["out_hw"] == (in_h, in_w) and in_h % 16 == 0 and in_w % 16 == 0):
                    # dedicated kernel for the network tail: LDS halo tiles, weights in registers
                    st0 = producers[0]
                    plan.steps = [st for st in plan.steps if st not in producers]
                    plan.steps.append(TailStep(org["name"] + "+" + p.node.name, org["srcs"][0].tensor, org["srcs"][1].tensor,
                                               org["srcs"][0].w, org["srcs"][1].w, st0.scale, st0.shift, head,
                                               in_h, in_w, org["macs"]))
                    ts.kind = "unused"
                    plan.layer_tensor = {k: v for k, v in plan.layer_tensor.items() if v != t_id}
                elif fusable:
                    # the head rides in the epilogue of the conv(s) producing its input; that
                    # tensor is never written (fp32 values go straight into the 1x1 contraction)
                    for st in producers:
                        st.head, st.out = head, -1
                    ts.kind = "unused"
                    plan.layer_tensor = {k: v for k, v in plan.layer_tensor.items() if v != t_id}
                else:
                    plan.steps.append(head)
                plan.classes = p.cout
                views[n.name] = _View(src.H, src.W)
        elif n.op == "maxpool":
            if n.attrs["pool"][0] != n.attrs["pool"][1] or n.attrs["strides"][0] != n.attrs["strides"][1]:
                raise PlanError(f"{n.name}: non-square pooling unsupported")
            pv = views[n.inputs[0]]
            pp = pv.pending
            pre = None
            if (pp is not None and pp.raw_needed and pp.residual < 0 and pp.stage == "relu" and consumers.get(n.inputs[0], 0) == 1
                    and pp.emitted_out < 0 and np.all(pp.raw_scale != 0)):
                # the conv's un-normalised output is stored anyway (a skip connection wants it): store
                # ONLY that, and let the pool apply BN + ReLU on the fly (saves one tensor write + read)
                a = pp.scale / pp.raw_scale
                pre = ((a).astype(np.float32), (pp.shift - pp.raw_shift * a).astype(np.float32))
                pp.scale, pp.shift, pp.relu = pp.raw_scale.copy(), pp.raw_shift.copy(), False
                pp.raw_needed = False
                emit(pp)                                         # single output == the raw tensor
                pp.emitted_raw = pp.emitted_out
                src_t = pp.emitted_out
                pv.pending = None
                plan.tensors[src_t].name += ":raw"
            else:
                src_t = plain_tensor(n.inputs[0])
            oh, ow, c = n.out_shape
            dst = new_tensor(oh, ow, c, n.name)
            plan.steps.append(PoolStep(n.name, src_t, dst, n.attrs["pool"][0], n.attrs["strides"][0],
                                       pre_scale=pre[0] if pre else None, pre_shift=pre[1] if pre else None,
                                       pre_relu=pre is not None))
            views[n.name] = _View(oh, ow, [Seg(dst, c)])
            plan.layer_tensor[n.name] = dst
        elif n.op == "convT":
            # Keras Conv2DTranspose (kernel [kh][kw][out][in]); lowered to output-parity classes in emit()
            src = gatherable(n.inputs[0])
            kh, kw = n.attrs["kernel"]
            sy, sx = n.attrs["strides"]
            oh, ow, cout = n.out_shape
            if (sy, sx) != (2, 2) or kh < 2 or kw < 2:
                raise PlanError(f"{n.name}: only stride-2 Conv2DTranspose with kernel >= 2 is lowered")
            if any(s.shift or s.tensor < 0 for s in src.segs) or len(src.segs) > 2:
                raise PlanError(f"{n.name}: Conv2DTranspose source must be one or two stored tensors")
            if n.attrs.get("activation", "linear") not in ("linear", "relu"):
                raise PlanError(f"{n.name}: inline activation {n.attrs['activation']} unsupported")
            # TF: the transposed conv is the gradient of a SAME/VALID conv with this kernel and stride
            pt = max(kh - sy, 0) // 2 if n.attrs["padding"] == "same" else 0
            pl = max(kw - sx, 0) // 2 if n.attrs["padding"] == "same" else 0
            wt = np.ascontiguousarray(np.transpose(weights[f"{n.name}/kernel:0"].astype(np.float32), (0, 1, 3, 2)))   # -> [kh][kw][in][out]
            bias = weights[f"{n.name}/bias:0"] if n.attrs["use_bias"] else None
            p = _Pending(n, src.segs, kh, kw, 1, 1, 0, 0, cout, wt, bias, (oh, ow), 0)
            p.convT, p.in_hw = (kh, kw, pt, pl), (src.H, src.W)
            if n.attrs.get("activation", "linear") == "relu":
                p.relu, p.stage = True, "relu"
            views[n.name] = _View(oh, ow, pending=p)
        else:
            raise PlanError(f"{n.name}: op {n.op} not lowered")

    if plan.classes == 0:
        raise PlanError("graph does not end in Conv2D 1x1 -> BatchNormalization -> softmax")
    for s in plan.steps:
        if s.kind == "conv":
            for seg in s.srcs:
                if seg.tensor < 0:
                    raise PlanError(f"{s.name}: unresolved network-input source")
    return plan
