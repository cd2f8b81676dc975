"""Times the pieces of bench.py's pipeline3 step (BASELINE configs[2]) one by one."""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd import _capi
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.stages import scaled_size
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
H, W = 3500, 2500
Hs, Ws = scaled_size(H, W)
page = synthetic_page(H, W, seed=0)
mb = SegModel(*calibrated_model(2, 448, 448, seed=11), max_batch=1, precision=prec)
ml = SegModel(*calibrated_model(4, 448, 448, seed=12), max_batch=108, precision=prec)
mt = SegModel(*calibrated_model(2, 448, 448, seed=0), max_batch=280, precision=prec)
if os.environ.get("PROBE_NULL_STREAM") == "1":              # as bench.py runs its handles: on torch's current (the legacy default) stream
    for m_ in (mb, ml, mt):
        m_.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    print("handles on the legacy default stream")
d_page = torch.from_numpy(page).cuda()
_, box, px = mb.ctx.extract_page_box(page, Hs, Ws)
print("box", box, px)
bw, bh = box[2], box[3]
d_a = torch.empty((bh, bw), dtype=torch.uint8, device="cuda"); d_b = torch.empty_like(d_a); d_c = torch.empty_like(d_a)

def t(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms")

t("extract_page_box (host page in)", lambda: mb.ctx.extract_page_box(page, Hs, Ws))
t("  segment_whole_scaled only", lambda: mb.ctx.segment_whole_scaled(page, Hs, Ws, Hs, Ws))
d_mask = torch.from_numpy(np.ascontiguousarray(mb.ctx.segment_whole_scaled(page, Hs, Ws, Hs, Ws))).cuda()
t("  page_box_dev only", lambda: mb.ctx.page_box_dev(d_mask.data_ptr(), Hs, Ws))
t("layout: segment_crop_dev (binarise)", lambda: ml.ctx.segment_crop_dev(d_page.data_ptr(), H, W, Hs, Ws, box, True, d_a.data_ptr()))
t("morph erode3 + dilate4", lambda: (ml.ctx.morph_dev(d_a.data_ptr(), bh, bw, 0, 5, 3, d_b.data_ptr()), ml.ctx.morph_dev(d_b.data_ptr(), bh, bw, 1, 5, 4, d_b.data_ptr())))
t("textline: segment_crop_dev", lambda: mt.ctx.segment_crop_dev(d_page.data_ptr(), H, W, Hs, Ws, box, False, d_c.data_ptr()))
d_up = torch.from_numpy(np.ascontiguousarray(np.zeros((Hs, Ws, 3), np.uint8))).cuda(); d_l = torch.empty((Hs, Ws), dtype=torch.uint8, device="cuda")
t("textline: segment_page_dev on 4200x3000", lambda: mt.ctx.segment_page_dev(d_up.data_ptr(), Hs, Ws, d_l.data_ptr()))
