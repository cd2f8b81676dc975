mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "layer or predict_448 or exact or transpose or full_page" > gpurun_out/pytest_gpu_r03s.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03s.log
for prec in f16x3 f16; do
ARGS="--precision $prec --no-cpu-baseline --no-second-mode --no-extras --steps 8 --warmup 2 --repeats 2"
for o in 1 0; do
SBBSEG_TAP_ORDER=$o SBBSEG_BENCH_OPS=gpurun_out/ops_r03s_${prec}_o$o.json timeout 600 python bench.py $ARGS > gpurun_out/bench_r03s_${prec}_o$o.log 2>&1
tail -1 gpurun_out/bench_r03s_${prec}_o$o.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec taporder=$o BENCH', d['value'], d['repeats']['patches_per_s'])"
done
python - <<PY
import json
vs=[1,0]
d={v:json.load(open(f'gpurun_out/ops_r03s_${prec}_o{v}.json')) for v in vs}
print('sums', {v: round(sum(o['ms_per_launch'] for o in d[v]),3) for v in vs})
for i,o in enumerate(d[1]):
    if 'conv2x2' in o['name']: print(f"{o['name']:48s}", ' '.join(f"{d[v][i]['ms_per_launch']:8.4f}" for v in vs))
PY
done
