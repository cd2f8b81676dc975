#!/bin/bash
# One GPU-box session: parity tests, bench (+per-op table), batch sweep, rocprofv3 kernel stats.
# usage: tools/gpu_round.sh <tag> [quick]
TAG=${1:-r}
mkdir -p gpurun_out
if [ "$2" != "quick" ]; then
  timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu_$TAG.log 2>&1
  echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log
else
  timeout 600 python -m pytest tests -m gpu -q -s -k "layer or predict_448 or segment_page or famil" > gpurun_out/pytest_gpu_$TAG.log 2>&1
  echo "pytest(quick) rc=$?"; tail -4 gpurun_out/pytest_gpu_$TAG.log
fi
SBBSEG_BENCH_OPS=gpurun_out/ops_$TAG.json timeout 600 python bench.py > gpurun_out/bench_$TAG.log 2>&1
tail -1 gpurun_out/bench_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BENCH', d['value'], 'patches/s', d['achieved_tflops_end_to_end'], 'TF e2e; roofline', d['roofline']['kernel'], d['roofline']['achieved'], 'all convs', d['roofline']['all_convs'], 'label', d['label_match'])"
for v in ${VARIANTS:-1}; do
  SBBSEG_BENCH_OPS=gpurun_out/ops_${TAG}_v$v.json timeout 300 python bench.py --conv-variant $v --no-cpu-baseline --steps 10 > gpurun_out/bench_${TAG}_v$v.log 2>&1
  tail -1 gpurun_out/bench_${TAG}_v$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv-variant $v', d['value'], 'patches/s', d['roofline']['all_convs'])"
done
for mb in 70; do
  timeout 300 python bench.py --max-batch $mb --no-cpu-baseline --steps 10 > gpurun_out/bench_${TAG}_mb$mb.log 2>&1
  tail -1 gpurun_out/bench_${TAG}_mb$mb.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_batch', d['config']['max_batch'], d['value'], 'patches/s')"
done
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -2
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -12 "$F"
