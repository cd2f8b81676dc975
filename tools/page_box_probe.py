import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.stages import scaled_size
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model
H, W = 3500, 2500
Hs, Ws = scaled_size(H, W)
page = synthetic_page(H, W, seed=0)
mb = SegModel(*calibrated_model(2, 448, 448, seed=11), max_batch=1, precision="f16x3")
d_mask = torch.from_numpy(np.ascontiguousarray(mb.ctx.segment_whole_scaled(page, Hs, Ws, Hs, Ws))).cuda()
for _ in range(3): mb.ctx.page_box_dev(d_mask.data_ptr(), Hs, Ws)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): mb.ctx.page_box_dev(d_mask.data_ptr(), Hs, Ws)
torch.cuda.synchronize(); print("page_box_dev %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
