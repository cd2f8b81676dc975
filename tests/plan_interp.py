"""numpy interpreter of a fused ``planner.Plan`` (test helper, CPU only).

Emulates the *semantics* libsbbseg gives each step (per-source gather with upsampling shift /
placement offset / zero padding / stride, folded scale+shift, residual, ReLU, strided output
placement, fused head, PAIRS/C8 input forms) so that the planner's lowering -- including the
parity split of the decoder convs -- can be checked against the unfused oracle without a GPU.
fp32 throughout; convolution arithmetic is delegated to the oracle's C conv.  Never used by the product."""
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


def source_conv(plan, vals, g, out_h, out_w):
    """Contribution of one source to the op's output grid, fp32 [n,out_h,out_w,cout]."""
    a = vals[g.tensor][..., :g.channels]
    if g.shift:
        a = np.repeat(np.repeat(a, 2, axis=1), 2, axis=2)
    n, h, w, c = a.shape
    # logical tensor: stored data placed at (off_y, off_x), zero elsewhere, unbounded to the right/bottom
    need_h = (out_h - 1) * g.stride_y - g.pad_top + g.kh
    need_w = (out_w - 1) * g.stride_x - g.pad_left + g.kw
    LH, LW = max(need_h, h + g.off_y), max(need_w, w + g.off_x)
    logical = np.zeros((n, LH, LW, c), np.float32)
    logical[:, g.off_y:g.off_y + h, g.off_x:g.off_x + w] = a
    xin = np.pad(logical, ((0, 0), (g.pad_top, 0), (g.pad_left, 0), (0, 0)))
    y = kf.conv2d(xin, g.w, None, (g.stride_y, g.stride_x), "valid")
    return y[:, :out_h, :out_w]


def run_plan(plan, x):
    """Returns (labels uint8 [n,H,W], probs float32 [n,H,W,C], vals)."""
    x = np.asarray(x, np.float32)
    n = x.shape[0]
    vals = make_input_forms(plan, x)
    labels = probs = None

    def place(tid, z, s):
        t = plan.tensors[tid]
        if tid not in vals:
            vals[tid] = np.zeros((n, t.H, t.W, t.C), np.float32)
        (sy, sx), (oy, ox) = s.out_stride, s.out_off
        vals[tid][:, oy::sy, ox::sx][:, :s.out_h, :s.out_w] = z

    for s in plan.steps:
        if s.kind == "conv":
            y = sum(source_conv(plan, vals, g, s.out_h, s.out_w) for g in s.srcs)
            if s.raw_out >= 0:
                place(s.raw_out, (y * s.raw_scale + s.raw_shift).astype(np.float32), s)
            z = y * s.scale + s.shift
            if s.residual >= 0:
                (sy, sx), (oy, ox) = s.out_stride, s.out_off
                z = z + vals[s.residual][:, oy::sy, ox::sx][:, :s.out_h, :s.out_w]
            if s.relu:
                z = np.maximum(z, 0)
            z = z.astype(np.float32)
            if s.out >= 0:
                place(s.out, z, s)
            if s.head is not None:
                hd = s.head
                pr = kf._softmax(((z @ hd.w) * hd.scale + hd.shift).astype(np.float32))
                if probs is None:
                    probs = np.zeros((n, plan.in_h, plan.in_w, hd.classes), np.float32)
                (sy, sx), (oy, ox) = s.out_stride, s.out_off
                probs[:, oy::sy, ox::sx][:, :s.out_h, :s.out_w] = pr
        elif s.kind == "tail":
            up = np.repeat(np.repeat(vals[s.src0][..., :64], 2, axis=1), 2, axis=2)
            xin = np.pad(np.concatenate([up, vals[s.img][..., :3]], axis=3), ((0, 0), (1, 1), (1, 1), (0, 0)))
            w = np.concatenate([s.w_src0, s.w_img], axis=2)
            z = np.maximum(kf.conv2d(xin, w, None, (1, 1), "valid") * s.scale + s.shift, 0).astype(np.float32)
            probs = kf._softmax(((z @ s.head.w) * s.head.scale + s.head.shift).astype(np.float32))
        elif s.kind == "maxpool":
            xin = vals[s.src]
            if s.pre_scale is not None:
                xin = xin * s.pre_scale + s.pre_shift
                if s.pre_relu:
                    xin = np.maximum(xin, 0)
            vals[s.dst] = kf._maxpool(xin.astype(np.float32), (s.k, s.k), (s.stride, s.stride))
        elif s.kind == "head":
            logits = (vals[s.src] @ s.w) * s.scale + s.shift
            probs = kf._softmax(logits.astype(np.float32))
    labels = np.argmax(probs, axis=3).astype(np.uint8)
    return labels, probs, vals
