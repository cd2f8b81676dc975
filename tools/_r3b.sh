mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_r03b.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu_r03b.log
SBBSEG_BENCH_OPS=gpurun_out/ops_r03b_x3.json timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-mode --no-extras > gpurun_out/bench_r03b_x3.log 2>&1
tail -1 gpurun_out/bench_r03b_x3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03b_x3.json'))
print('sum ms', sum(o['ms_per_launch'] for o in d))
for o in d:
    if any(k in o['name'] for k in ('tail','direct','conv2x2')): print(o['name'], o['ms_per_launch'])
PY
SBBSEG_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch-pages 8 --steps 2 --warmup 1 --repeats 1 > gpurun_out/bench_r03b_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -c 2500 gpurun_out/bench_r03b_gloo2.log
