mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_x3" > gpurun_out/pytest_gpu_r03d.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03d.log
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras"
SBBSEG_BENCH_OPS=gpurun_out/ops_r03d_x3.json timeout 600 python bench.py --steps 10 --warmup 3 $ARGS > gpurun_out/bench_r03d_x3.log 2>&1
tail -1 gpurun_out/bench_r03d_x3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
python - <<PY
import json
d=json.load(open('gpurun_out/ops_r03d_x3.json'))
print('sum ms', sum(o['ms_per_launch'] for o in d))
for o in d[:8]: print(o['name'], o['ms_per_launch'])
PY
BENCH_ARGS="$ARGS" bash tools/pmc_run.sh pmc_r03d > gpurun_out/pmc_run_r03d.log 2>&1
python tools/pmc_report.py pmc_r03d gpurun_out/ops_r03d_x3.json gpurun_out/pmc_summary_r03d.json f16x3 140 > gpurun_out/pmc_per_op_r03d.txt 2>&1; head -12 gpurun_out/pmc_per_op_r03d.txt; tail -3 gpurun_out/pmc_per_op_r03d.txt
rm -rf gpurun_out/pmc_r03d_sq gpurun_out/pmc_r03d_fetch gpurun_out/pmc_r03d_write
