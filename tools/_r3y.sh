mkdir -p gpurun_out
tools/probes/bin/block_probe
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused or predict_448 or block" > gpurun_out/pytest_gpu_r03y.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03y.log
ARGS="--no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
SBBSEG_BENCH_OPS=gpurun_out/ops_r03y.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03y.log 2>&1
tail -1 gpurun_out/bench_r03y.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
a=json.load(open('gpurun_out/ops_r03y.json'))
print('sum', round(sum(o['ms_per_launch'] for o in a),3), [(o['name'][:20], round(o['ms_per_launch'],3)) for o in a if 'block' in o['name']])
PY
