mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or predict_448 or fused or exact or tail or whole" > gpurun_out/pytest_gpu_r03h.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03h.log
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 8 --warmup 2 --repeats 2"
for v in 0 1048576; do
SBBSEG_BENCH_OPS=gpurun_out/ops_r03h_v$v.json timeout 600 python bench.py $ARGS --conv-variant $v > gpurun_out/bench_r03h_v$v.log 2>&1
tail -1 gpurun_out/bench_r03h_v$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v BENCH', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
vs=[0,1048576]
d={v:json.load(open(f'gpurun_out/ops_r03h_v{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
for i,o in enumerate(d[0]):
    if any(k in o['name'] for k in ('block','direct','tail')): print(f"{o['name']:48s}", ' '.join(f"{d[v][i]['ms_per_launch']:8.4f}" for v in vs))
PY
