#!/bin/bash
mkdir -p gpurun_out/s7
for ns in 0 3 4; do SBBSEG_X3_SMALL_NS=$ns PROBE_SIZES=1,2,4,8,16,35 python tools/small_batch_probe.py f16x3 ops 2>&1 | grep -v "Warn\|synthetic_w\|amdgpu.ids" > gpurun_out/s7/small_ns$ns.txt; done
SBBSEG_X3_SMALL_NS=4 python -m pytest tests/test_gpu_parity.py -x -q -k "every_fused_layer or predict_448 or whole_image or batch_one" > gpurun_out/s7/tests_ns4.txt 2>&1
head -8 gpurun_out/s7/small_ns*.txt; tail -3 gpurun_out/s7/tests_ns4.txt
paste <(grep " us" gpurun_out/s7/small_ns0.txt | head -50) <(grep " us" gpurun_out/s7/small_ns4.txt | head -50 | awk '{print $2, $3}')
