bash tools/collect_profiles.sh r03a f16x3
# the N > 1 code path (batch64, sharded pages, all-gather, per-rank rates) on this 1-GPU box: two ranks share the GPU, gloo
SBBSEG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch-pages 8 --steps 2 --warmup 1 --repeats 1 > gpurun_out/bench_r03a_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -c 1500 gpurun_out/bench_r03a_gloo2.log
