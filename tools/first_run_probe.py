"""GPU box: does the FIRST pooled run of a fresh handle give the same label maps as the second?  (round 6: a first-run-only difference on one
page of four in the plain fp16 mode; this probe repeats the scenario under the environment it is started with.)
usage: [SBBSEG_C3ER=0] [SBBSEG_OWNED_REGIONS=0] python tools/first_run_probe.py [precision] [handles]"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sbb_textline_detection_amd.model import SegModel  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402
from tools.synth_model import calibrated_model  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
n_handles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg, w = calibrated_model(2, 448, 448, seed=0)
pages = [torch.from_numpy(synthetic_page(3500, 2500, seed=70 + k)).cuda() for k in range(4)]
bad = 0
for h in range(n_handles):
    m = SegModel(cfg, w, device=0, max_batch=280, precision=prec)
    outs = [torch.empty((3500, 2500), dtype=torch.uint8, device="cuda") for _ in pages]
    runs = []
    for rep in range(3):
        for o in outs:
            o.fill_(7)
        m.ctx.segment_pages_dev([p_.data_ptr() for p_ in pages], 3500, 2500, [o.data_ptr() for o in outs])
        torch.cuda.synchronize()
        runs.append([o.cpu().numpy().copy() for o in outs])
    for rep in (0, 1):
        for k in range(4):
            d = runs[rep][k] != runs[2][k]
            if d.any():
                bad += 1
                ys, xs = np.nonzero(d)
                print(f"handle {h} run {rep} page {k}: {int(d.sum())} labels differ from run 2; rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; "
                      f"tile rows {sorted(set((ys // 360).tolist()))[:12]} tile cols {sorted(set((xs // 360).tolist()))[:12]}")
    m.release()
print("env", {k: v for k, v in os.environ.items() if k.startswith("SBBSEG_")}, prec, "differences:", bad)
