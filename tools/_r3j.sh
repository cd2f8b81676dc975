mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or predict_448 or tail or exact or whole or transpose or segment_page" > gpurun_out/pytest_gpu_r03j.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03j.log
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras --steps 8 --warmup 2 --repeats 2"
for w in 1 0; do
SBBSEG_TAIL_X3_W8=$w SBBSEG_BENCH_OPS=gpurun_out/ops_r03j_w$w.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03j_w$w.log 2>&1
tail -1 gpurun_out/bench_r03j_w$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8=$w BENCH', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
vs=[1,0]
d={v:json.load(open(f'gpurun_out/ops_r03j_w{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
for i,o in enumerate(d[1]):
    if any(k in o['name'] for k in ('tail',)): print(f"{o['name']:48s}", ' '.join(f"{d[v][i]['ms_per_launch']:8.4f}" for v in vs))
PY
