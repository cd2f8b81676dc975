"""Per-layer table from a rocprofv3 --kernel-trace CSV: the last launch of every plan op (the bench's roofline pass: one
whole chunk per launch on one lane), joined with the op names / FLOP counts bench.py wrote (SBBSEG_BENCH_OPS), so the
rocprof durations can be read beside the HIP-event durations of the same launches.
usage: tools/trace_per_op.py <kernel_trace.csv> <ops.json>"""
import csv
import json
import sys

PLAN = ("conv_igemm", "maxpool", "head_kernel", "dec_tail", "stem_conv", "stem_pool", "dec_halo", "expand_reduce", "conv3x3_c64_direct", "bottleneck", "block_x3")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if any(s in r["Kernel_Name"] for s in PLAN):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ops = [o for o in json.load(open(sys.argv[2])) if o.get("ms_per_launch", 1.0) >= 0.02]      # (ops that launch nothing have no dispatch)
last = rows[-len(ops):]
print("| # | op | kernel | rocprof us | HIP-event us | algorithmic TFLOP/s (rocprof) | issued TFLOP/s (rocprof) |")
print("|---|---|---|---|---|---|---|")
tot_r = tot_e = 0.0
for i, (o, (t0, t1, kn)) in enumerate(zip(ops, last)):
    us = (t1 - t0) / 1e3
    ev = o["ms_per_launch"] * 1e3
    short = kn.split("(")[0].replace("void sbbseg::", "")
    if len(short) > 60:
        short = short[:57] + "..."
    scale = ev / us if us else 0
    print("| %d | %s | `%s` | %.1f | %.1f | %.0f | %.0f |" % (i, o["name"], short, us, ev, o["tflops"] * scale, o["tflops_issued"] * scale))
    tot_r += us
    tot_e += ev
print("| | sum | | %.1f | %.1f | | |" % (tot_r, tot_e))
