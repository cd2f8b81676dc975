mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or predict_448 or exact or fused or full_page" > gpurun_out/pytest_gpu_r03r.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03r.log
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 8 --warmup 2 --repeats 2"
SBBSEG_BENCH_OPS=gpurun_out/ops_r03r_x3.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03r_x3.log 2>&1
tail -1 gpurun_out/bench_r03r_x3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03r_x3.json'))
print('sum ms', round(sum(o['ms_per_launch'] for o in d),3))
seen=set()
for o in d:
    if o['name'] in seen: continue
    seen.add(o['name'])
    if any(k in o['name'] for k in ('block','conv2x2','tail','111')): print(f"{o['name']:48s} {o['ms_per_launch']:8.4f}")
PY
