mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -s -x -k "ingest_forms or every_fused_layer" > gpurun_out/r2a_layers.log 2>&1; echo "layers rc=$?"; tail -15 gpurun_out/r2a_layers.log
timeout 600 python tests/diag_cmp448.py > gpurun_out/r2a_cmp448.log 2>&1; echo "cmp rc=$?"; tail -8 gpurun_out/r2a_cmp448.log
SBBSEG_BENCH_OPS=gpurun_out/ops_r2a_x3.json timeout 600 python bench.py --precision f16x3 --steps 10 > gpurun_out/bench_r2a_x3.log 2>&1; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_r2a_x3.log
