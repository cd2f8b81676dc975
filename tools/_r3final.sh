bash tools/collect_profiles.sh r03 f16x3
bash tools/collect_profiles.sh r03 f16 skip-tests
SBBSEG_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --batch-pages 8 --steps 2 --warmup 1 --repeats 1 > gpurun_out/bench_r03_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -1 gpurun_out/bench_r03_gloo2.log > gpurun_out/bench_r03_gloo2.json; python -c "
import json; d=json.load(open('gpurun_out/bench_r03_gloo2.json')); print('gloo2', d['value'], d['config']['workload'][:60], d['exchange'], d['per_rank'], d['ranks_seen']['world_size'])"
