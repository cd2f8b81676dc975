"""Wall time of the three-stage wrapper sequence (InferenceStages.run) on one page, host arrays in and out."""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbb_textline_detection_amd import stages
from sbb_textline_detection_amd.weights import save_sbbw
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model
d = tempfile.mkdtemp()
specs = {"model_page_mixed_best": 2, "model_strukturerkennung": 4, "model_textline_new": 2}
for name, classes in specs.items():
    cfg, w = calibrated_model(classes, 448, 448, seed=classes)
    save_sbbw(os.path.join(d, name + ".sbbw"), cfg, w)
st = stages.InferenceStages(*[os.path.join(d, n + ".h5") for n in specs], model_kwargs={"max_batch": 108})
page = synthetic_page(3500, 2500, seed=1)
for k in range(4):
    t0 = time.perf_counter()
    st.get_image_and_scales(page)
    t1 = time.perf_counter(); mask = st.extract_page_mask()
    t2 = time.perf_counter(); reg = st.extract_text_regions()
    t3 = time.perf_counter(); lines = st.textline_contours()
    t4 = time.perf_counter()
    print("run %d: border %.1f ms, layout %.1f ms, textline %.1f ms, total %.1f ms (%d forwards)" %
          (k, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, 1 + 2 * 108), flush=True)
print(mask.shape, reg.shape, lines.shape)
