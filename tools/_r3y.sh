mkdir -p gpurun_out
for prec in f16 f16x3; do
ARGS="--precision $prec --no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
for v in 0 1 0 1; do
SBBSEG_SPLIT_ISSUE_ALL=$v SBBSEG_BENCH_OPS=gpurun_out/ops_r03y_${prec}_$v.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03y_$v.log 2>&1
tail -1 gpurun_out/bench_r03y_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH $prec split_all=$v', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
a=json.load(open('gpurun_out/ops_r03y_${prec}_1.json')); b=json.load(open('gpurun_out/ops_r03y_${prec}_0.json'))
print('sum', round(sum(o['ms_per_launch'] for o in a),3), round(sum(o['ms_per_launch'] for o in b),3))
for x,y in zip(a,b):
    if abs(x['ms_per_launch']-y['ms_per_launch'])>0.01*y['ms_per_launch'] and 'conv2x2' in x['name'] or 'conv3x3' in x['name']: print(f"{x['name']:46s} {x['ms_per_launch']:.3f} {y['ms_per_launch']:.3f}")
PY
done
