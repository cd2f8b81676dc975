"""Manufacture a *calibrated* synthetic model (stand-in for a trained ``.h5``; none is available).

Plain seeded weights (``weights.synthetic_weights``) give a net whose BatchNorm statistics do not
match its activations: residual sums then grow geometrically and the softmax saturates into a
constant label map -- useless as a parity workload.  A trained model's moving_mean /
moving_variance *do* match its activations, so here they are set that way: one fp64 torch-CPU pass
over a small seeded calibration batch, each BatchNormalization's moving statistics replaced by the
statistics of its own input (then gamma/beta keep their seeded jitter).

This is weight *manufacturing* only -- it is neither the product inference path (HIP, via
libsbbseg) nor the oracle (oracle/), and results are only ever compared within one process.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from sbb_textline_detection_amd.keras_graph import parse_model_config, resnet50_unet_config  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402
from sbb_textline_detection_amd.weights import save_sbbw, synthetic_weights  # noqa: E402


def forward_torch(graph, weights, x_nhwc, dtype=torch.float32, calibrate_bn=False):
    """fp32/fp64 forward of a parsed Keras graph with torch-CPU ops (NCHW inside).
    With ``calibrate_bn`` the BN moving statistics in ``weights`` are overwritten in place by the
    batch statistics of each BN's input before it is applied."""
    vals = {}
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    for n in graph.nodes:
        a = [vals[i] for i in n.inputs]
        if n.op == "input":
            y = t(x_nhwc).permute(0, 3, 1, 2).contiguous()
        elif n.op == "zeropad":
            tp, b, l, r = n.attrs["pad"]
            y = F.pad(a[0], (l, r, tp, b))
        elif n.op == "conv":
            w = t(weights[f"{n.name}/kernel:0"]).permute(3, 2, 0, 1).contiguous()
            bias = t(weights[f"{n.name}/bias:0"]) if n.attrs["use_bias"] else None
            xin = a[0]
            if n.attrs["padding"] == "same":
                kh, kw = n.attrs["kernel"]; sy, sx = n.attrs["strides"]
                H, W = xin.shape[2:]
                th = max((-(-H // sy) - 1) * sy + kh - H, 0); tw = max((-(-W // sx) - 1) * sx + kw - W, 0)
                xin = F.pad(xin, (tw // 2, tw - tw // 2, th // 2, th - th // 2))
            y = F.conv2d(xin, w, bias, stride=n.attrs["strides"])
        elif n.op == "convT":
            # Keras kernel [kh][kw][out][in] -> torch conv_transpose2d weight [in][out][kh][kw]
            w = t(weights[f"{n.name}/kernel:0"]).permute(3, 2, 0, 1).contiguous()
            bias = t(weights[f"{n.name}/bias:0"]) if n.attrs["use_bias"] else None
            kh, kw = n.attrs["kernel"]; sy, sx = n.attrs["strides"]
            full = F.conv_transpose2d(a[0], w, bias, stride=(sy, sx))                 # (H-1)*s + k rows: the 'valid' result for k >= s
            H, W = a[0].shape[2:]
            if n.attrs["padding"] == "same":
                pt, pl = max(kh - sy, 0) // 2, max(kw - sx, 0) // 2
                full = F.pad(full, (0, max(pl + W * sx - full.shape[3], 0), 0, max(pt + H * sy - full.shape[2], 0)))
                y = full[:, :, pt:pt + H * sy, pl:pl + W * sx]
            else:
                y = F.pad(full, (0, max(W * sx - full.shape[3], 0), 0, max(H * sy - full.shape[2], 0)))
        elif n.op == "bn":
            if calibrate_bn:
                m = a[0].mean(dim=(0, 2, 3)); v = a[0].var(dim=(0, 2, 3), unbiased=False)
                weights[f"{n.name}/moving_mean:0"] = m.to(torch.float32).numpy().copy()
                weights[f"{n.name}/moving_variance:0"] = np.maximum(v.to(torch.float32).numpy(), 1e-4).copy()
            y = F.batch_norm(a[0], t(weights[f"{n.name}/moving_mean:0"]), t(weights[f"{n.name}/moving_variance:0"]),
                             t(weights[f"{n.name}/gamma:0"]) if n.attrs.get("scale", True) else None,
                             t(weights[f"{n.name}/beta:0"]) if n.attrs.get("center", True) else None, False, 0.0, n.attrs["eps"])
        elif n.op == "act":
            k = n.attrs["kind"]
            y = F.relu(a[0]) if k == "relu" else F.softmax(a[0], dim=1) if k == "softmax" else a[0]
        elif n.op == "maxpool":
            y = F.max_pool2d(a[0], n.attrs["pool"], n.attrs["strides"])
        elif n.op == "upsample":
            y = F.interpolate(a[0], scale_factor=n.attrs["size"], mode="nearest")
        elif n.op == "concat":
            y = torch.cat(a, dim=1)
        elif n.op == "add":
            y = a[0] + a[1]
        elif n.op == "crop_last":
            y = a[0][:, :, :-1, :-1]
        else:
            raise NotImplementedError(n.op)
        vals[n.name] = y
    return vals[graph.output_name].permute(0, 2, 3, 1).contiguous().to(torch.float32).numpy()


def _make_decisive(graph, w, strength=6.0):
    """Give the seeded net a trained-like, *decisive* output: strengthen the direct image path of the
    last decoder conv (its concat takes the network input) into a few channels and let the head read
    them, so that logit differences follow ink vs. paper (bimodal) instead of being noise around 0.
    The deep path still contributes; BN statistics are calibrated afterwards as usual."""
    byn = graph.by_name()
    node = byn[graph.output_name]
    chain = []
    while node.op != "concat":
        if node.op == "conv":
            chain.append(node)
        node = byn[node.inputs[0]]
    head, tail = chain[0], chain[1]
    kt = w[f"{tail.name}/kernel:0"]                      # [3][3][C_up + 3][32]; the image channels come last
    kh = w[f"{head.name}/kernel:0"]                      # [1][1][32][classes]
    for j in range(8):
        sgn = 1.0 if j % 2 == 0 else -1.0
        kt[1, 1, -3:, j] += sgn * strength / 3.0
        for c in range(kh.shape[3]):
            kh[0, 0, j, c] += sgn * (2.0 if c % 2 == 0 else -2.0)


def calibrated_model(n_classes=2, height=448, width=448, seed=0, calib_hw=160, calib_batch=2, decisive=False):
    """(model_config, weights): seeded weights with BN statistics calibrated on synthetic pages.
    ``decisive``: see :func:`_make_decisive` (a trained net's outputs are decisive; plain random
    weights give a worst-case, noise-like label map where every pixel is a potential near-tie)."""
    cfg = resnet50_unet_config(n_classes, height, width)
    w = synthetic_weights(parse_model_config(cfg), seed)
    if decisive:
        _make_decisive(parse_model_config(cfg), w)
    cal_cfg = resnet50_unet_config(n_classes, calib_hw, calib_hw)
    cal_graph = parse_model_config(cal_cfg)
    page = synthetic_page(calib_hw * 2, calib_hw * calib_batch, seed=seed + 1000)
    xs = np.stack([page[calib_hw // 2:calib_hw // 2 + calib_hw, i * calib_hw:(i + 1) * calib_hw]
                   for i in range(calib_batch)]).astype(np.float64) / 255.0
    nt = torch.get_num_threads()
    torch.set_num_threads(1)               # fixed summation order -> same weights on every machine
    try:
        forward_torch(cal_graph, w, xs, torch.float64, calibrate_bn=True)
    finally:
        torch.set_num_threads(nt)
    return cfg, w


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a calibrated synthetic .sbbw model")
    ap.add_argument("out"); ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--size", type=int, default=448); ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    cfg, w = calibrated_model(a.classes, a.size, a.size, a.seed)
    save_sbbw(a.out, cfg, w)
    print("wrote", a.out)
