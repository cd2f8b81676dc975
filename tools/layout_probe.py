"""Why is the layout stage of configs[2] slower than the textline stage on the same 108 tiles?  Per-op HIP-event times of both
handles on the same crop (profiling runs one lane), with and without the Otsu binarisation."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.stages import scaled_size
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
H, W = 3500, 2500
Hs, Ws = scaled_size(H, W)
page = synthetic_page(H, W, seed=0)
d_page = torch.from_numpy(page).cuda()
box = (0, 0, Ws, Hs)
d_a = torch.empty((Hs, Ws), dtype=torch.uint8, device="cuda")
models = {"layout(4 classes, seed 12)": SegModel(*calibrated_model(4, 448, 448, seed=12), max_batch=108, precision=prec),
          "textline(2 classes, seed 0)": SegModel(*calibrated_model(2, 448, 448, seed=0), max_batch=108, precision=prec),
          "2 classes, seed 12": SegModel(*calibrated_model(2, 448, 448, seed=12), max_batch=108, precision=prec)}


def wall(m, binar, n=5):
    f = lambda: m.ctx.segment_crop_dev(d_page.data_ptr(), H, W, Hs, Ws, box, binar, d_a.data_ptr())
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


prof = {}
for name, m in models.items():
    for binar in (False, True):
        ms = wall(m, binar)
        m.ctx.profile_enable(True); m.ctx.profile_reset()
        for _ in range(3):
            m.ctx.segment_crop_dev(d_page.data_ptr(), H, W, Hs, Ws, box, binar, d_a.data_ptr())
        torch.cuda.synchronize()
        rows = m.ctx.profile()
        m.ctx.profile_enable(False)
        tot = sum(r["total_ms"] for r in rows) / 3
        prof[(name, binar)] = {r["name"]: r["total_ms"] / 3 for r in rows if r["launches"]}
        print(f"{name:30s} binarise={binar!s:5s}: wall {ms:7.2f} ms (two lanes), per-op sum on one lane {tot:7.2f} ms")
keys = list(prof)
names = list(prof[keys[0]])
print("per-op ms (one lane of 108):", " | ".join(f"{k[0][:10]}/{'bin' if k[1] else 'raw'}" for k in keys))
for i, n in enumerate(names):
    vals = []
    for k in keys:
        ks = list(prof[k])
        vals.append(prof[k][ks[i]] if i < len(ks) else float("nan"))
    if max(vals) - min(vals) > 0.03:
        print(f"  {n:48s}" + " ".join(f"{v:7.3f}" for v in vals))
