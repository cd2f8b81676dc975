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
def t(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, lab = t(lambda: m.ctx.segment_page(page)); print("ctx.segment_page (C ABI, host in/out): %.1f ms" % ms)
ms, _ = t(lambda: m.segment_page(page)); print("SegModel.segment_page: %.1f ms" % ms)
ms, out = t(lambda: predict.do_prediction(True, page, m)); print("do_prediction: %.1f ms" % ms)
ms, _ = t(lambda: np.repeat(lab[:, :, None], 3, axis=2)); print("np.repeat x3: %.1f ms" % ms)
ms, _ = t(lambda: np.ascontiguousarray(page, np.uint8)); print("ascontiguousarray(page): %.2f ms" % ms)
import torch
tp = torch.from_numpy(page)
ms, _ = t(lambda: (tp.cuda(), torch.cuda.synchronize())); print("torch pageable H2D 26 MB: %.1f ms" % ms)
pp = tp.pin_memory()
ms, _ = t(lambda: (pp.cuda(non_blocking=True), torch.cuda.synchronize())); print("torch pinned H2D 26 MB: %.1f ms" % ms)
d = torch.empty((3500, 2500), dtype=torch.uint8, device="cuda")
ms, _ = t(lambda: d.cpu()); print("torch D2H 8.75 MB (pageable): %.1f ms" % ms)
