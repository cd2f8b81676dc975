mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "predict_448 or segment_page or layer or famil" > gpurun_out/pytest_gpu_r03y.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03y.log
ARGS="--precision f16 --no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
for v in 1 0 1 0; do
SBBSEG_X3_SPLIT_ISSUE=$v SBBSEG_BENCH_OPS=gpurun_out/ops_r03y_$v.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03y_$v.log 2>&1
tail -1 gpurun_out/bench_r03y_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH f16 split_issue=$v', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
a=json.load(open('gpurun_out/ops_r03y_1.json')); b=json.load(open('gpurun_out/ops_r03y_0.json'))
print('sum', round(sum(o['ms_per_launch'] for o in a),3), round(sum(o['ms_per_launch'] for o in b),3))
for x,y in zip(a,b):
    if abs(x['ms_per_launch']-y['ms_per_launch'])>0.008: print(f"{x['name']:46s} {x['ms_per_launch']:.3f} {y['ms_per_launch']:.3f}")
PY
