mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-second-mode --repeats 1 --steps 8 --warmup 2 $EXTRA > gpurun_out/bench_$tag.log 2>&1; tail -1 gpurun_out/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', d['value'], 'p/s; dom', r['kernel'], r['achieved'], r['frac'], 'issued', r['frac_issued'], 'k3', r['conv3x3_stages']['frac'])"; }
EXTRA="--pages-per-step 16" run b70l2 A=1
EXTRA="--pages-per-step 16" run b70l1 SBBSEG_LANES=1
EXTRA="--pages-per-step 4 --max-batch 280" run b280l1 SBBSEG_LANES=1 SBBSEG_BENCH_PAGE=7000x5000 SBBSEG_BENCH_OPS=gpurun_out/ops_b280l1.json
EXTRA="--pages-per-step 4 --max-batch 280" run b280l2 SBBSEG_BENCH_PAGE=7000x5000
EXTRA="--pages-per-step 4 --max-batch 140" run b140l2 SBBSEG_BENCH_PAGE=7000x5000
EXTRA="--pages-per-step 4 --max-batch 140" run b140l1 SBBSEG_LANES=1 SBBSEG_BENCH_PAGE=7000x5000
