mkdir -p gpurun_out
SBBSEG_BENCH_OPS=gpurun_out/ops_r2d.json timeout 900 python bench.py > gpurun_out/bench_r2d.log 2>&1; echo "bench rc=$?"; tail -c 4500 gpurun_out/bench_r2d.log
