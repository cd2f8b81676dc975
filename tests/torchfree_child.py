"""Child process of test_run_is_torch_free_at_full_size: InferenceStages.run() in an interpreter where PyTorch CANNOT be imported
(the reference's environment is Keras/TF: requirements.txt has no torch).  argv: <dir with a/ notext/ smallbox/ model sets> <out.npz>.
Runs every scenario twice -- page resident in library-owned device buffers (the default) and stage by stage through the host entry
points (SBBSEG_STAGES_RESIDENT=0) -- and checks the two agree; the resident results of scenario `a` go to out.npz for the parent's
oracle check.  Prints one JSON line."""
import importlib.abc
import json
import os
import sys


class _NoTorch(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name == "torch" or name.startswith("torch."):
            raise ImportError("torch is deliberately unimportable in this process")
        return None


sys.meta_path.insert(0, _NoTorch())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402

from sbb_textline_detection_amd import clear_session, stages  # noqa: E402
from sbb_textline_detection_amd.synthetic import synthetic_page  # noqa: E402

NAMES = ("model_page_mixed_best", "model_strukturerkennung", "model_textline_new")          # main.py:58-60


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    page = synthetic_page(3500, 2500, seed=33)
    summary = {}
    for scen in ("a", "notext", "smallbox"):
        st = stages.InferenceStages(*[os.path.join(root, scen, n + ".h5") for n in NAMES], model_kwargs={"max_batch": 108})
        os.environ["SBBSEG_STAGES_RESIDENT"] = "1"
        res = st.run(page)
        box, thr = st.page_box, getattr(st, "otsu_threshold", None)
        st.get_image_and_scales(page)
        assert st._run_resident() is not None, "the resident path does not apply"
        os.environ["SBBSEG_STAGES_RESIDENT"] = "0"
        ref = st.run(page)
        assert box == st.page_box and res[3] == ref[3], (scen, box, st.page_box)
        if res[1] is not None:
            assert thr == st.otsu_threshold
        for x, y in zip(res[:3], ref[:3]):
            assert (x is None) == (y is None), scen
            if x is not None:
                assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x, y), scen
        summary[scen] = {"box": list(box), "regions": res[1] is not None, "textlines": res[2] is not None,
                         "class1_pixels": int((res[1][:, :, 0] == 1).sum()) if res[1] is not None else -1}
        if scen == "a":
            np.savez_compressed(out_path, mask=res[0][:, :, 0], regions=res[1][:, :, 0], lines=res[2], box=np.array(box), thr=np.int64(thr))
        clear_session()
    summary["torch_loaded"] = any(m == "torch" or m.startswith("torch.") for m in sys.modules)
    print("TORCHFREE " + json.dumps(summary))


if __name__ == "__main__":
    main()
