#!/bin/bash
# A/B per-op times of conv-variant settings: tools/ab_ops.sh "<variant> <variant> ..."   (GPU box)
mkdir -p gpurun_out
for v in $1; do
  SBBSEG_BENCH_OPS=gpurun_out/ops_v$v.json python bench.py --no-cpu-baseline --conv-variant $v 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], 'patches/s', d['ms_per_step'], 'ms')"
done
python - "$@" <<'PY'
import json, sys
vs = sys.argv[1].split()
tabs = [json.load(open("gpurun_out/ops_v%s.json" % v)) for v in vs]
print("%-48s" % "op", *["v%-8s" % v for v in vs])
for i, o in enumerate(tabs[0]):
    print("%-48s" % o["name"][:48], *["%.4f   " % t[i]["ms_per_launch"] for t in tabs])
print("%-48s" % "sum", *["%.4f   " % sum(o["ms_per_launch"] for o in t) for t in tabs])
PY
