mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -x -k "famil or every_fused_layer or predict_448_matches or repeatab" > gpurun_out/r2e_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r2e_pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-second-mode --repeats 1 --steps 8 --warmup 2 $EXTRA > gpurun_out/bench_$tag.log 2>&1; tail -1 gpurun_out/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], 'p/s; dom', r['kernel'], r['achieved'], r['frac'], 'issued', r['frac_issued'], 'k3', r['conv3x3_stages']['frac'])"; }
EXTRA="" run xr SBBSEG_BENCH_OPS=gpurun_out/ops_xr.json
EXTRA="--conv-variant 131072" run noxr SBBSEG_BENCH_OPS=gpurun_out/ops_noxr.json
