#!/bin/bash
# A/B of environment knobs on the GPU box: tools/env_ab.sh "NAME=VAL;NAME2=VAL2" ...   (each argument = one configuration; "-" = no knob)
# prints the timed-region rate and the per-op table (one-lane profiling pass) of every configuration side by side.
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs=$(echo "$cfg" | tr ';' ' '); [ "$cfg" = "-" ] && envs=""
  env $envs SBBSEG_BENCH_OPS=gpurun_out/ops_ab$i.json python bench.py --no-cpu-baseline --no-second-mode --no-extras --steps 12 --warmup 3 --repeats 2 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$i] $cfg:', d['value'], d['repeats']['patches_per_s'], 'label', (d.get('label_match') or {}).get('label_mismatches'))"
done
python - "$@" <<'PY'
import json, sys
n = len(sys.argv) - 1
tabs = [json.load(open("gpurun_out/ops_ab%d.json" % (i + 1))) for i in range(n)]
print("%-48s" % "op", *["[%d]      " % (i + 1) for i in range(n)])
for i, o in enumerate(tabs[0]):
    print("%-48s" % o["name"][:48], *["%.4f   " % t[i]["ms_per_launch"] for t in tabs])
print("%-48s" % "sum", *["%.4f   " % sum(o["ms_per_launch"] for o in t) for t in tabs])
PY
