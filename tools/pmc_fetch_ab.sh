#!/bin/bash
# FETCH_SIZE per plan op (one counter per pass: FETCH_SIZE + WRITE_SIZE together exceed the hardware and hang the run)
# FETCH_SIZE per plan op for several conv-variant settings: tools/pmc_fetch_ab.sh "<v> <v> ..."   (GPU box)
REPO=$(pwd); mkdir -p gpurun_out
for v in $1; do
  (cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/fab_$v -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --conv-variant $v > $REPO/gpurun_out/fab_$v.log 2>&1)
done
SBBSEG_BENCH_OPS=gpurun_out/ops_names.json python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
python - "$1" <<'PY'
import csv, json, sys
from collections import defaultdict
vs = sys.argv[1].split()
names = [o["name"] for o in json.load(open("gpurun_out/ops_names.json"))]
cols = []
for v in vs:
    rows = defaultdict(dict); kn = {}
    for r in csv.DictReader(open("gpurun_out/fab_%s/pmc_counter_collection.csv" % v)):
        d = int(r["Dispatch_Id"]); rows[d][r["Counter_Name"]] = float(r["Counter_Value"]); kn[d] = r["Kernel_Name"]
    ids = [d for d in sorted(kn) if any(s in kn[d] for s in ("conv_igemm", "maxpool", "head_kernel", "dec_tail", "stem_conv", "conv3x3_c64_direct"))][-len(names):]
    cols.append([rows[d].get("FETCH_SIZE", 0) * 2 / 1024 for d in ids])
print("%-46s" % "op (fetch MB, x2-corrected)", *["v%-14s" % v for v in vs])
for i, n in enumerate(names):
    print("%-46s" % n[:46], *["%9.1f      " % c[i] for c in cols])
PY
