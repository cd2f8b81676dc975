"""Seam-2 latency: model.predict on one 448x448 patch (what the unmodified reference loop calls 70-108 x per page)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from sbb_textline_detection_amd import predict
from tools.synth_model import calibrated_model

cfg, w = calibrated_model(2, 448, 448, seed=0)
m = SegModel(cfg, w, device=0, max_batch=70, precision="f16")
page = synthetic_page(3500, 2500, seed=0)
x = (page[:448, :448][None] / 255.0)
for _ in range(3): m.predict(x)
t0 = time.perf_counter()
for _ in range(50): p = m.predict(x)
dt = (time.perf_counter() - t0) / 50
print("predict(n=1): %.2f ms per call = %.0f patches/s" % (dt * 1e3, 1 / dt))
x8 = np.repeat(x, 8, axis=0)
for _ in range(2): m.predict(x8)
t0 = time.perf_counter()
for _ in range(10): p = m.predict(x8)
dt = (time.perf_counter() - t0) / 10
print("predict(n=8): %.2f ms per call = %.0f patches/s" % (dt * 1e3, 8 / dt))
# the reference-style Python loop over the page through seam 2: one predict per tile, tiling on the host
# (a float page is not eligible for the fused path, so do_prediction takes its host loop; batch_size=1 = main.py:287)
t0 = time.perf_counter()
out = predict.do_prediction(True, page.astype(np.float64), m, batch_size=1)
dt = time.perf_counter() - t0
print("reference-style loop (host tiling + SegModel.predict per tile): %.2f s per page = %.0f patches/s" % (dt, 70 / dt))
t0 = time.perf_counter()
out2 = predict.do_prediction(True, page, m)
dt = time.perf_counter() - t0
print("fused do_prediction (host page in, host labels out): %.1f ms per page = %.0f patches/s" % (dt * 1e3, 70 / dt))
print("label agreement loop vs fused: %.6f" % (out[:, :, 0] == out2[:, :, 0]).mean())
