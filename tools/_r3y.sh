mkdir -p gpurun_out
ARGS="--no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
for v in 1024 100000000 1024 100000000; do
SBBSEG_X3_T256_MINK=$v SBBSEG_BENCH_OPS=gpurun_out/ops_r03y_$v.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03y_$v.log 2>&1
tail -1 gpurun_out/bench_r03y_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH t256_mink=$v', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
a=json.load(open('gpurun_out/ops_r03y_100000000.json')); b=json.load(open('gpurun_out/ops_r03y_1024.json'))
print('sum', round(sum(o['ms_per_launch'] for o in a),3), round(sum(o['ms_per_launch'] for o in b),3))
for x,y in zip(a,b):
    if abs(x['ms_per_launch']-y['ms_per_launch'])>0.015*y['ms_per_launch']: print(f"{x['name']:46s} {x['ms_per_launch']:.3f} {y['ms_per_launch']:.3f}")
PY
