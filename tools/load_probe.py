"""Where does the first-use latency of a model go?  (load .sbbw -> parse -> plan -> pack/upload)"""
import sys, os, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from sbb_textline_detection_amd import _capi, model as M
from sbb_textline_detection_amd.weights import save_sbbw, load_sbbw, synthetic_model
from sbb_textline_detection_amd.keras_graph import parse_model_config
from sbb_textline_detection_amd.planner import build_plan
t1 = time.perf_counter(); print("imports %.2f s" % (t1 - t0))
cfg, w = synthetic_model(2, 448, 448, 0)
d = tempfile.mkdtemp(); path = os.path.join(d, "m.sbbw"); save_sbbw(path, cfg, w)
t0 = time.perf_counter(); cfg2, w2 = load_sbbw(path); t1 = time.perf_counter(); print("load_sbbw %.3f s" % (t1 - t0))
g = parse_model_config(cfg2); t2 = time.perf_counter()
plan = build_plan(g, w2); t3 = time.perf_counter(); print("build_plan %.3f s" % (t3 - t2))
ctx = _capi.Context(0, _capi.PREC_F16); t4 = time.perf_counter(); print("context create (HIP init) %.3f s" % (t4 - t3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
ctx.load_plan(plan, 70); t5 = time.perf_counter(); print("load_plan (pack + upload + alloc) %.3f s" % (t5 - t4))
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(6)
ctx.close()
t0 = time.perf_counter(); m = M.SegModel(cfg2, w2, max_batch=70); t1 = time.perf_counter(); print("SegModel() second construction %.3f s" % (t1 - t0))
