mkdir -p gpurun_out
ARGS="--no-cpu-baseline --no-second-mode --no-extras --steps 6 --warmup 2 --repeats 2"
for v in 256 128; do
SBBSEG_X3_BC64_TILE=$v SBBSEG_BENCH_OPS=gpurun_out/ops_r03y_$v.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03y_$v.log 2>&1
tail -1 gpurun_out/bench_r03y_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH bc64 tile=$v', d['value'], d['repeats']['patches_per_s'], d.get('label_match'))"
done
python - <<PY
import json
a=json.load(open('gpurun_out/ops_r03y_128.json')); b=json.load(open('gpurun_out/ops_r03y_256.json'))
print('sum', round(sum(o['ms_per_launch'] for o in a),3), round(sum(o['ms_per_launch'] for o in b),3))
for x,y in zip(a,b):
    if 'c192to64' in x['name']: print(f"{x['name']:46s} {x['ms_per_launch']:.3f} {y['ms_per_launch']:.3f}")
PY
