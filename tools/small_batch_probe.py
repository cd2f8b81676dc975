"""Device time of n tiles through one handle, n = 1 .. max_batch (page resident in HBM, labels left on the device):
where small launches lose against the pooled 140-tile lanes, and -- with `ops` -- which ops carry a one-patch forward."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
show_ops = len(sys.argv) > 2 and sys.argv[2] == "ops"
variant = int(os.environ.get("PROBE_VARIANT", "0"), 0)
sizes = [1, 2, 4, 8, 16, 24, 35, 54, 70, 108, 140, 216, 280] if not variant else [1, 2, 4, 8, 16, 35]
if os.environ.get("PROBE_SIZES"):
    sizes = [int(v) for v in os.environ["PROBE_SIZES"].split(",")]
m = SegModel(*calibrated_model(2, 448, 448, seed=0), max_batch=280, precision=prec)
H, W = 4000 * 3, 3000                       # 324 tiles
page = synthetic_page(H, W, seed=0)
d_page = torch.from_numpy(page).cuda()
d_out = torch.empty((324, 448, 448), dtype=torch.uint8, device="cuda")
ctx = m.ctx
if os.environ.get("PROBE_LANES"):
    ctx.set_lanes(int(os.environ["PROBE_LANES"]))
    print("lanes", os.environ["PROBE_LANES"])
if os.environ.get("PROBE_OWNED"):
    ctx.set_owned_regions(int(os.environ["PROBE_OWNED"]))      # 2: the tile-range entry point computes owned regions too
    print("owned regions mode", os.environ["PROBE_OWNED"])
if variant:
    ctx.set_conv_variant(variant)
    print(f"conv variant {variant:#x}")
print(f"{prec}: tiles per call -> ms per call, patches/s  (device-resident page, two lanes from 16 tiles up)")
for n in sizes:
    reps = max(3, min(50, 2000 // (n * 10 + 20)))
    for _ in range(2):
        ctx.segment_tile_range_dev(d_page.data_ptr(), H, W, 0, n, d_out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.segment_tile_range_dev(d_page.data_ptr(), H, W, 0, n, d_out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"  n = {n:4d}: {dt * 1e3:8.3f} ms  {n / dt:8.0f} patches/s  ({dt / n * 1e6:7.1f} us per patch)")
if show_ops:
    for n in (1, 8):
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(10):
            ctx.segment_tile_range_dev(d_page.data_ptr(), H, W, 0, n, d_out.data_ptr())
        torch.cuda.synchronize()
        rows = ctx.profile()
        ctx.profile_enable(False)
        tot = sum(r["total_ms"] for r in rows) / 10
        print(f"per-op HIP-event times at n = {n} (sum {tot:.3f} ms):")
        for r in rows:
            if r["launches"]:
                print(f"    {r['name']:48s} {r['total_ms'] / r['launches'] * 1e3:8.1f} us")
