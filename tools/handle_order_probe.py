"""Does the two-lane overlap depend on WHICH handle of a process runs?  k identical handles, the same 108-tile crop through each
(wall time of the two-lane run; the per-op sums on one lane are equal).  Run with GPU_MAX_HW_QUEUES=4 (ROCm's default) / 8 / 16."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = 4200, 3000
page = synthetic_page(H, W, seed=0)
d_page = torch.from_numpy(page).cuda()
d_a = torch.empty((H, W), dtype=torch.uint8, device="cuda")
cfg, w = calibrated_model(2, 448, 448, seed=0)
ms = [SegModel(cfg, w, max_batch=108, precision=prec) for _ in range(k)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), " SBBSEG_LANE_PRIORITY =", os.environ.get("SBBSEG_LANE_PRIORITY"))
for rnd in range(2):
    for i, m in enumerate(ms):
        f = lambda: m.ctx.segment_page_dev(d_page.data_ptr(), H, W, d_a.data_ptr())
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        print(f"round {rnd} handle {i}: {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms per 108-tile page")
