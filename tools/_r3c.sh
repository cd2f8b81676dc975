mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_x3 or fused_bottleneck or predict_448 or layer" > gpurun_out/pytest_gpu_r03c.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_r03c.log
SBBSEG_BENCH_OPS=gpurun_out/ops_r03c_x3.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-mode --no-extras > gpurun_out/bench_r03c_x3.log 2>&1
tail -1 gpurun_out/bench_r03c_x3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03c_x3.json'))
print('sum ms', sum(o['ms_per_launch'] for o in d))
for o in d[:14]: print(o['name'], o['ms_per_launch'])
PY
