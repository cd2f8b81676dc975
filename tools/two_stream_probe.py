"""Probe: does splitting a page's tiles over two handles on two streams fill the launch tails?"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbb_textline_detection_amd import _capi
from sbb_textline_detection_amd.model import SegModel
from sbb_textline_detection_amd.synthetic import synthetic_page
from tools.synth_model import calibrated_model

cfg, w = calibrated_model(2, 448, 448, seed=0)
page = synthetic_page(3500, 2500, seed=0)
d_page = torch.from_numpy(page).cuda()
n = _capi.tile_grid(3500, 2500, 448, 448)[0].shape[0]
d_tiles = torch.empty((n, 448, 448), dtype=torch.uint8, device="cuda")

def bench(split, steps=20):
    parts = [(i * n // split, (i + 1) * n // split) for i in range(split)]
    models = [SegModel(cfg, w, device=0, max_batch=max(b - a for a, b in parts), precision="f16") for _ in parts]
    streams = [torch.cuda.Stream() for _ in parts]
    for m, s in zip(models, streams):
        m.ctx.set_stream(s.cuda_stream)
    def step():
        for m, (a, b) in zip(models, parts):
            m.ctx.segment_tile_range_dev(d_page.data_ptr(), 3500, 2500, a, b - a, d_tiles[a:].data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("split", split, "ms/page %.3f" % ms, "patches/s %.0f" % (n / ms * 1e3), flush=True)
    for m in models: m.release()

def bench_pages(k, steps=20):
    models = [SegModel(cfg, w, device=0, max_batch=n, precision="f16") for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    outs = [torch.empty((n, 448, 448), dtype=torch.uint8, device="cuda") for _ in range(k)]
    for m, s in zip(models, streams):
        m.ctx.set_stream(s.cuda_stream)
    def step():
        for m, o in zip(models, outs):
            m.ctx.segment_tile_range_dev(d_page.data_ptr(), 3500, 2500, 0, n, o.data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("pages in flight", k, "ms/step %.3f" % ms, "patches/s %.0f" % (k * n / ms * 1e3), flush=True)
    for m in models: m.release()

def bench_pages_offset(frac, steps=20):
    """two whole pages in flight on two handles (one lane each), the second stream started `frac` of a page later"""
    os.environ["SBBSEG_LANES"] = "1"
    models = [SegModel(cfg, w, device=0, max_batch=n, precision="f16") for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.empty((n, 448, 448), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for m, s in zip(models, streams):
        m.ctx.set_stream(s.cuda_stream)
    def run(steps):
        if frac > 0:
            models[1].ctx.segment_tile_range_dev(d_page.data_ptr(), 3500, 2500, 0, max(1, int(n * frac)), outs[1].data_ptr())
        for _ in range(steps):
            for m, o in zip(models, outs):
                m.ctx.segment_tile_range_dev(d_page.data_ptr(), 3500, 2500, 0, n, o.data_ptr())
    run(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("2 pages in flight, offset %.2f page: ms/step %.3f" % (frac, ms), "patches/s %.0f" % ((2 * n * steps + (n * frac if frac else 0)) / (ms * steps) * 1e3), flush=True)
    for m in models: m.release()
    os.environ.pop("SBBSEG_LANES")

bench(2)
for f in (0.0, 0.25, 0.5):
    bench_pages_offset(f)
