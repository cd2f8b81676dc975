"""Where does the two-lane timed region lose time BETWEEN kernels?  From a rocprofv3 --kernel-trace CSV of a short bench run:
per hardware queue the busy time and the gaps between consecutive kernels, and over both lanes the time during which 0 / 1 / 2
plan kernels were running (union of the intervals).
usage: tools/timeline_gaps.py <kernel_trace.csv> [first_fraction last_fraction]   (the slice of the run to analyse, default 0.5 0.9:
the timed steps; the roofline pass at the end runs one lane)"""
import csv
import sys
from collections import defaultdict

PLAN = ("conv_igemm", "maxpool", "head_kernel", "dec_tail", "stem_conv", "stem_pool", "dec_halo", "expand_reduce", "conv3x3_c64_direct", "bottleneck", "block_x3")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-70:],
                     any(s in r["Kernel_Name"] for s in PLAN)))
rows.sort()
if not rows:
    sys.exit(f"{sys.argv[1]}: empty kernel trace")
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * f0
t_hi = rows[0][0] + (rows[-1][1] - rows[0][0]) * f1
sel = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
if not sel:
    sys.exit(f"{sys.argv[1]}: no kernel inside the [{f0}, {f1}] slice of the trace ({len(rows)} rows in all)")
span = sel[-1][1] - sel[0][0]
print(f"{len(sel)} kernels in a {span / 1e6:.2f} ms slice; {sum(1 for r in sel if r[4])} of them plan ops")
byq = defaultdict(list)
for r in sel:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r[1] - r[0] for r in rs)
    gaps = [b[0] - a[1] for a, b in zip(rs, rs[1:])]
    pos = sorted(g for g in gaps if g > 0)
    med = pos[len(pos) // 2] / 1e3 if pos else 0
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms = {busy / span:.3f} of the slice; gaps between consecutive kernels: "
          f"sum {sum(pos) / 1e6:.2f} ms, median {med:.1f} us, > 50 us: {sum(1 for g in pos if g > 50e3)}, overlapping starts: {sum(1 for g in gaps if g <= 0)}")
# union: how many kernels run at once
ev = []
for r in sel:
    ev.append((r[0], 1)); ev.append((r[1], -1))
ev.sort()
depth, last, hist = 0, ev[0][0], defaultdict(int)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
for k in sorted(hist):
    print(f"  {k} kernel(s) running: {hist[k] / 1e6:8.2f} ms = {hist[k] / span:.3f}")
# the longest gaps on each queue, with their neighbours
for q, rs in sorted(byq.items()):
    gl = sorted(((b[0] - a[1], a[3], b[3]) for a, b in zip(rs, rs[1:])), reverse=True)[:5]
    for g, a, b in gl:
        print(f"  queue {q}: gap {g / 1e3:8.1f} us after {a[-40:]} before {b[-40:]}")
