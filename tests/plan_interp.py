"""numpy interpreter of a fused ``planner.Plan`` (test helper, CPU only).

Emulates the *semantics* libsbbseg gives each step (gather with upsampling shift / placement offset
/ zero padding, folded scale+shift, residual, ReLU, PAIRS/C8 input forms) so that the planner's
lowering can be checked against the unfused oracle without a GPU.  fp32 throughout; convolution
arithmetic is delegated to the oracle's C conv.  Never used by the product."""
import numpy as np

from oracle import keras_forward as kf


def make_input_forms(plan, x):
    """x: float32 [n,H,W,3] -> {tensor id: array} for the plan's input-form tensors."""
    n, H, W, _ = x.shape
    out = {}
    for tid, t in enumerate(plan.tensors):
        if t.kind == "input_c8":
            a = np.zeros((n, H, W, 8), np.float32)
            a[..., :3] = x
            out[tid] = a
        elif t.kind == "input_pairs":
            p = t.pad
            padded = np.zeros((n, H + 2 * p, 2 * t.W, 4), np.float32)
            padded[:, p:p + H, p:p + W, :3] = x
            out[tid] = padded.reshape(n, H + 2 * p, t.W, 8)
    return out


def logical_source(plan, vals, seg, LH, LW):
    a = vals[seg.tensor][..., :seg.channels]
    if seg.shift:
        a = np.repeat(np.repeat(a, 2, axis=1), 2, axis=2)
    n, h, w, c = a.shape
    out = np.zeros((n, LH, LW, c), np.float32)
    hh, ww = min(h, LH - seg.off_y), min(w, LW - seg.off_x)
    out[:, seg.off_y:seg.off_y + hh, seg.off_x:seg.off_x + ww] = a[:, :hh, :ww]
    return out


def run_plan(plan, x):
    """Returns (labels uint8 [n,H,W], probs float32 [n,H,W,C], vals)."""
    vals = make_input_forms(plan, np.asarray(x, np.float32))
    labels = probs = None
    for s in plan.steps:
        if s.kind == "conv":
            ot = plan.tensors[s.out if s.out >= 0 else s.raw_out]
            # logical input extent needed by the last window
            LH = (ot.H - 1) * s.stride_y - s.pad_top + s.kh
            LW = (ot.W - 1) * s.stride_x - s.pad_left + s.kw
            LH = max([LH] + [(plan.tensors[g.tensor].H << g.shift) + g.off_y for g in s.srcs])
            LW = max([LW] + [(plan.tensors[g.tensor].W << g.shift) + g.off_x for g in s.srcs])
            xin = np.concatenate([logical_source(plan, vals, g, LH, LW) for g in s.srcs], axis=3)
            xin = np.pad(xin, ((0, 0), (s.pad_top, s.kh), (s.pad_left, s.kw), (0, 0)))
            y = kf.conv2d(xin, s.w_hwio, None, (s.stride_y, s.stride_x), "valid")[:, :ot.H, :ot.W]
            if s.raw_out >= 0:
                vals[s.raw_out] = (y * s.raw_scale + s.raw_shift).astype(np.float32)
            if s.out >= 0:
                z = y * s.scale + s.shift
                if s.residual >= 0:
                    z = z + vals[s.residual]
                if s.relu:
                    z = np.maximum(z, 0)
                vals[s.out] = z.astype(np.float32)
        elif s.kind == "maxpool":
            vals[s.dst] = kf._maxpool(vals[s.src], (s.k, s.k), (s.stride, s.stride))
        elif s.kind == "head":
            logits = (vals[s.src] @ s.w) * s.scale + s.shift
            probs = kf._softmax(logits.astype(np.float32))
            labels = np.argmax(probs, axis=3).astype(np.uint8)
    return labels, probs, vals
