mkdir -p gpurun_out
ARGS="--no-cpu-baseline --no-second-mode --no-extras --steps 8 --warmup 2 --repeats 2"
run() { tag=$1; shift; timeout 900 python bench.py $ARGS "$@" > gpurun_out/bench_r03x_$tag.log 2>&1; tail -1 gpurun_out/bench_r03x_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH $tag', d['value'], d['repeats']['patches_per_s'], d['config'].get('max_batch'), d['config'].get('tiles_per_step'))" 2>&1 | tail -1; }
run mb280 --max-batch 280 --pages-per-step 20
run mb334 --max-batch 334 --pages-per-step 19
run mb250 --max-batch 250 --pages-per-step 25
run mb280b --max-batch 280 --pages-per-step 20
