"""Compare the three arithmetic modes against the oracle on 448x448 patches (run on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from gpu_common import compare_probs, make_model, patches_from_page  # noqa: E402
from oracle import keras_forward as kf  # noqa: E402

x = (patches_from_page(448, 448, 2, seed=9) / 255.0).astype(np.float32)
ref = None
for prec in ("f32", "f16x3", "bf16", "f16"):
    cfg, w, g, model = make_model(2, 448, 448, seed=2, precision=prec, max_batch=4)
    if ref is None:
        t = time.time(); ref = kf.forward(g, w, x)
        print("oracle s/patch", (time.time() - t) / 2, "threads", kf.num_threads())
    got = model.predict(x)
    d, mism, bad = compare_probs(ref, got, 0.06)
    srt = np.sort(ref, -1); margin = srt[..., -1] - srt[..., -2]
    mm = ref.argmax(-1) != got.argmax(-1)
    print(prec, "max|dp| %.5f mean|dp| %.6f label mismatch frac %.6f outside tol %d, max margin among mismatches %.5f" % (
        d, float(np.abs(ref - got).mean()), mism / ref[..., 0].size, bad, float(margin[mm].max()) if mism else 0.0))
    model.release()
