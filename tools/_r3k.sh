mkdir -p gpurun_out
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 --repeats 1"
run() { tag=$1; shift; env "$@" SBBSEG_BENCH_OPS=gpurun_out/ops_r03k_$tag.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03k_$tag.log 2>&1; tail -1 gpurun_out/bench_r03k_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag BENCH', d['value'])"; }
run base X=1
run t512k1024 SBBSEG_X3_T512_MINK=1024
run t256k4096 SBBSEG_X3_T256_MINK=4096
run t256k99999 SBBSEG_X3_T256_MINK=99999 SBBSEG_X3_T512_MINK=1024
python - <<PY
import json
vs=['base','t512k1024','t256k4096','t256k99999']
d={v:json.load(open(f'gpurun_out/ops_r03k_{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
seen=set()
for i,o in enumerate(d['base']):
    row=[d[v][i]['ms_per_launch'] for v in vs]
    if o['name'] in seen: continue
    if max(row)-min(row) > 0.03*max(row): seen.add(o['name']); print(f"{o['name']:48s}", ' '.join(f'{x:8.4f}' for x in row))
PY
