"""GPU probe: is the whole-image branch (border model, one patch, split-K launches) repeatable, and do its host and device entry
points agree?  Builds a border model whose class-1 answer is a near-tie on a few pixels (the `smallbox` scenario of
tests/test_gpu_parity.py::test_run_is_torch_free_at_full_size) and runs sbbseg_extract_page_box / _dev repeatedly.
    python tools/border_repeat_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sbb_textline_detection_amd.keras_graph import parse_model_config  # noqa: E402
from sbb_textline_detection_amd.model import SegModel  # noqa: E402
from sbb_textline_detection_amd.predict import resize_nearest  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402
from tools.synth_model import calibrated_model  # noqa: E402


def main():
    cfg, w = calibrated_model(2, 448, 448, seed=21)
    page = synthetic_page(3500, 2500, seed=33)
    bn = [n.name for n in parse_model_config(cfg).nodes if n.op == "bn"][-1]
    m = SegModel(cfg, w, device=0, max_batch=1)
    x = resize_nearest(resize_nearest(page, 4200, 3000), 448, 448)[None].astype(np.float32) / np.float32(255.0)
    p = m.predict(x)[0].astype(np.float64)
    m.release()
    margin = np.sort((np.log(p[..., 1]) - np.log(p[..., 0])).reshape(-1))
    print("top margins", margin[-6:])
    w = dict(w)
    b = w[bn + "/beta:0"].copy()
    b[1] -= 0.5 * (margin[-3] + margin[-4])
    w[bn + "/beta:0"] = b
    for ksplit in (True, False):
        m = SegModel(cfg, w, device=0, max_batch=1)
        c = m.ctx
        c.set_ksplit(ksplit)
        d_page = c.device_alloc(page.size)
        c.upload(d_page, page)
        d_mask = c.device_alloc(4200 * 3000)
        ref_mask = None
        for it in range(8):
            mask_h, box_h, px_h = c.extract_page_box(page, 4200, 3000, channels=1)
            box_d, px_d = c.extract_page_box_dev(d_page, 3500, 2500, 4200, 3000, d_mask)
            mask_d = c.download_labels(d_mask, 4200, 3000, 1)
            if ref_mask is None:
                ref_mask = mask_h.copy()
            print(f"ksplit={ksplit} it={it} host box {box_h} px {px_h} ones {int(mask_h.sum())} | dev box {box_d} px {px_d} ones {int(mask_d.sum())} | "
                  f"host==first {np.array_equal(mask_h, ref_mask)} dev==host {np.array_equal(mask_d, mask_h)}")
        pr = m.predict(x)[0]
        print("predict(n=1) class-1 pixels:", int((pr.argmax(-1) == 1).sum()))
        c.device_free(d_page); c.device_free(d_mask)
        m.release()


if __name__ == "__main__":
    main()
