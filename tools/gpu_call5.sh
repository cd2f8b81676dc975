#!/bin/bash
# scratch: one gpurun call -- small-launch ring depth, timeline gaps of the timed region, pipeline3 pieces
mkdir -p gpurun_out/s5
for ns in 0 3 4; do SBBSEG_X3_SMALL_NS=$ns PROBE_SIZES=1,2,4,8,16 python tools/small_batch_probe.py f16x3 ops 2>&1 | grep -v "Warn\|synthetic_w\|amdgpu.ids" > gpurun_out/s5/small_ns$ns.txt; done
python tools/pipeline3_probe.py f16x3 2>&1 | grep -v "Warn\|synthetic_w\|amdgpu.ids" > gpurun_out/s5/p3.txt
REPO=$(pwd)
ARGS="--precision f16x3 --no-cpu-baseline --no-second-mode --no-extras"
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/s5/prof -o prof -- python $REPO/bench.py --steps 3 --warmup 2 --repeats 2 $ARGS > $REPO/gpurun_out/s5/rocprof.log 2>&1)
T=$(find gpurun_out/s5/prof -name "*kernel_trace.csv" | head -1)
for sl in "0.35 0.5" "0.5 0.65" "0.65 0.8"; do python tools/timeline_gaps.py "$T" $sl; done > gpurun_out/s5/gaps.txt 2>&1
head -3 "$T" > gpurun_out/s5/trace_head.txt
rm -rf gpurun_out/s5/prof
head -20 gpurun_out/s5/small_ns*.txt | grep "n = \|=="; cat gpurun_out/s5/p3.txt; cat gpurun_out/s5/gaps.txt | head -60
