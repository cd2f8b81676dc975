mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or predict_448 or tail or batch64 or whole or segment_page or famil" > gpurun_out/pytest_gpu_r03t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03t.log
ARGS="--precision f16 --no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
SBBSEG_BENCH_OPS=gpurun_out/ops_r03t_f16.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03t_f16.log 2>&1
tail -1 gpurun_out/bench_r03t_f16.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03t_f16.json'))
print('sum ms', round(sum(o['ms_per_launch'] for o in d),3))
for o in d:
    if any(k in o['name'] for k in ('tail','block','conv2x2')): print(f"{o['name']:48s} {o['ms_per_launch']:8.4f}")
PY
