#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes; no --sys/--hip trace with --pmc).
TAG=${1:-pmc}
REPO=$(pwd)
mkdir -p gpurun_out
run() { # name, counters...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 5 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $REPO/gpurun_out/${TAG}_$name -o pmc -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $REPO/gpurun_out/${TAG}_$name.log 2>&1)
  ls gpurun_out/${TAG}_$name | head -5
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
