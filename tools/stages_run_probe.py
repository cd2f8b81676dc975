import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
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
for mode in ("1", "0", "1", "0"):
    os.environ["SBBSEG_STAGES_RESIDENT"] = mode
    st.run(page)
    t0 = time.perf_counter()
    for _ in range(3):
        out = st.run(page)
    print("InferenceStages.run, resident=%s: %.1f ms per page (host page in, three host masks out)" % (mode, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
print([None if o is None else getattr(o, "shape", o) for o in out])
