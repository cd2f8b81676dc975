mkdir -p gpurun_out
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 --repeats 1"
for v in 0 1; do
SBBSEG_BENCH_OPS=gpurun_out/ops_r03g_v$v.json timeout 600 python bench.py $ARGS --conv-variant $v > gpurun_out/bench_r03g_v$v.log 2>&1
tail -1 gpurun_out/bench_r03g_v$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v BENCH', d['value'])"
done
python - <<PY
import json
vs=[0,1]
d={v:json.load(open(f'gpurun_out/ops_r03g_v{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
for i,o in enumerate(d[0]):
    row=[d[v][i]['ms_per_launch'] for v in vs]
    if max(row)-min(row) > 0.03*max(row): print(f"{o['name']:48s}", ' '.join(f'{x:8.4f}' for x in row))
PY
