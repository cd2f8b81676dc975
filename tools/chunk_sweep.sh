#!/bin/bash
# GPU box: timed-region throughput of the page workload against the chunk size (tiles per two-lane chunk) -- launches whose tile counts
# fill whole rounds of the persistent grids (lane batch x 196 px close to a multiple of 256 x 256-px tiles) lose nothing to partial rounds.
#   tools/chunk_sweep.sh "280:16 320:32 ..."      (max_batch:pages_per_step)
mkdir -p gpurun_out
for spec in $1; do
  mb=${spec%%:*}; pps=${spec##*:}
  steps=$(( 160 / pps )); [ $steps -lt 3 ] && steps=3
  SBBSEG_BENCH_OPS=gpurun_out/ops_mb$mb.json timeout 300 python bench.py --max-batch $mb --pages-per-step $pps --steps $steps --warmup 2 --repeats 2 --no-cpu-baseline --no-second-mode --no-extras 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('max_batch $mb pages/step $pps:', d['value'], 'patches/s', d['repeats']['patches_per_s'], '| roofline', r['kernel'][-40:], r['avg_launch_ms'], 'ms', r['frac'], '| k3', r['conv3x3_stages']['frac'])"
done
