"""GPU probe: time of the device Otsu (histogram + threshold) on a 4200 x 3000 page region, text-like and noise pages."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import stage_glue
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import noise_page, synthetic_page
from tools.synth_model import calibrated_model

cfg, w = calibrated_model(2, 64, 64, seed=0)
m = SegModel(cfg, w, device=0, max_batch=2)
c = m.ctx
for name, page in (("synthetic page", synthetic_page(4200, 3000, seed=3)), ("uniform noise", noise_page(4200, 3000, seed=4))):
    d = c.device_alloc(page.size); t = c.device_alloc(4)
    c.upload(d, page)
    c.otsu_dev(d, 4200, 3000, t); c.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        c.otsu_dev(d, 4200, 3000, t)
    c.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    thr = int(c.download(t, (1,), np.int32)[0])
    print(f"otsu_dev on a 4200 x 3000 {name}: {ms:.3f} ms, threshold {thr} (oracle {stage_glue.otsu_threshold(page[:, :, 0])})")
    c.device_free(d); c.device_free(t)
m.release()
