mkdir -p gpurun_out
tools/probes/bin/tail_probe 140 2 0
tools/probes/bin/tail_probe 140 2 0
SBBSEG_PRECISION=f16 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tail or predict_448 or segment_page or whole or famil" > gpurun_out/pytest_gpu_r03u.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03u.log
ARGS="--precision f16 --no-cpu-baseline --no-second-mode --no-extras --steps 10 --warmup 3 --repeats 2"
timeout 600 python bench.py $ARGS > gpurun_out/bench_r03u.log 2>&1
tail -1 gpurun_out/bench_r03u.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['dtype'], d['value'], d['repeats']['patches_per_s'])"
