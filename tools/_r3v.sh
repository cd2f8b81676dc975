mkdir -p gpurun_out
ARGS="--no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 --repeats 1"
for v in base t512 half; do
  case $v in base) E="";; t512) E="SBBSEG_X3_BC64_TILE=512";; half) E="SBBSEG_X3_BC64_HALFGRID=1";; esac
  env $E SBBSEG_BENCH_OPS=gpurun_out/ops_r03v_$v.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03v_$v.log 2>&1
  tail -1 gpurun_out/bench_r03v_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH $v', d['value'], d['label_match'])"
  python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03v_$v.json'))
print('  sum ms', round(sum(o['ms_per_launch'] for o in d),3), [ (o['name'][:22], round(o['ms_per_launch'],3)) for o in d if 'c192to64' in o['name'] or 'tail' in o['name']])
PY
done
