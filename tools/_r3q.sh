mkdir -p gpurun_out
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 --repeats 1"
run() { tag=$1; shift; env "$@" SBBSEG_BENCH_OPS=gpurun_out/ops_r03q_$tag.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03q_$tag.log 2>&1; tail -1 gpurun_out/bench_r03q_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag BENCH', d['value'])"; }
run base X=1
run fgsmall SBBSEG_FG_X3_SMALL=1
run lanes1 SBBSEG_LANES=1
run fgmin4 SBBSEG_FG_MIN=4
python - <<PY
import json
vs=['base','fgsmall','lanes1','fgmin4']
d={v:json.load(open(f'gpurun_out/ops_r03q_{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
seen=set()
for i,o in enumerate(d['base']):
    row=[d[v][i]['ms_per_launch'] for v in vs]
    if o['name'] in seen: continue
    if max(row)-min(row) > 0.03*max(row): seen.add(o['name']); print(f"{o['name']:48s}", ' '.join(f'{x:8.4f}' for x in row))
PY
