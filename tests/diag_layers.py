"""Layer-by-layer HIP-vs-oracle report (run on the GPU box; prints where errors first appear)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SBBSEG_FUSE_BLOCKS"] = "0"   # reads intermediate tensors
from gpu_common import make_model, patches_from_page  # noqa: E402
from oracle import keras_forward as kf  # noqa: E402


def main():
    hw = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
    for precision in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("f32", "f16x3", "bf16", "f16")):
        cfg, w, g, model = make_model(2, hw[0], hw[1], seed=2, precision=precision, max_batch=4, calib_hw=64)
        x = (patches_from_page(hw[0], hw[1], 3, seed=4) / 255.0).astype(np.float32)
        taps = {name: None for name in model.plan.layer_tensor}
        ref = kf.forward(g, w, x, taps=taps)
        got = model.predict(x)
        print(f"== {precision}: max|dsoftmax| {np.abs(ref - got).max():.5f}  label mismatch {(ref.argmax(-1) != got.argmax(-1)).mean():.5f}")
        for name, tid in model.plan.layer_tensor.items():
            t = model.plan.tensors[tid]
            a = model.ctx.debug_read_tensor(tid, 3, (t.H, t.W, t.C))
            r = taps[name]
            rel = np.abs(a - r).max() / (np.abs(r).max() + 1e-6)
            flag = "  <<<<" if rel > (1e-3 if precision == "f32" else 0.05) else ""
            print(f"   {name:24s} {str((t.H, t.W, t.C)):18s} rel {rel:.3e} refmax {np.abs(r).max():.3f}{flag}")
        model.release()


if __name__ == "__main__":
    main()
