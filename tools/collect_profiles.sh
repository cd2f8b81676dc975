# One GPU-box session: full GPU test suite, smoke, the default bench line, and for ONE arithmetic mode the rocprofv3 kernel
# stats + per-op trace table + three PMC passes.
# usage: bash tools/collect_profiles.sh <tag> [precision: f16x3 | f16] [skip-tests]
# (results under gpurun_out/; copy what should be judged into profiles/)
TAG=${1:-r06}
PREC=${2:-f16x3}
mkdir -p gpurun_out
if [ "$3" != "skip-tests" ]; then
  timeout -k 5 2700 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
  timeout -k 5 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log > gpurun_out/bench_$TAG.json
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("BENCH", d["dtype"], d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_issued"))
print("  modes", (d.get("roofline") or {}).get("modes"))
print("  extras", d.get("extras")); print("  cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
PY
fi
ARGS="--precision $PREC --no-cpu-baseline --no-second-mode --no-extras"
SBBSEG_BENCH_OPS=gpurun_out/ops_${TAG}_$PREC.json timeout -k 5 600 python bench.py --steps 5 --warmup 2 $ARGS > gpurun_out/bench_${TAG}_$PREC.log 2>&1
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 5 --warmup 2 $ARGS > $REPO/gpurun_out/rocprof_$TAG.log 2>&1)
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" gpurun_out/kernel_stats_${TAG}_$PREC.csv && head -8 "$F"
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python tools/trace_per_op.py "$T" gpurun_out/ops_${TAG}_$PREC.json > gpurun_out/kernel_trace_per_op_${TAG}_$PREC.md 2>&1; tail -5 gpurun_out/kernel_trace_per_op_${TAG}_$PREC.md
rm -rf gpurun_out/prof_$TAG
BENCH_ARGS="$ARGS" bash tools/pmc_run.sh pmc_$TAG > gpurun_out/pmc_run_$TAG.log 2>&1
python tools/pmc_report.py pmc_$TAG gpurun_out/ops_${TAG}_$PREC.json gpurun_out/pmc_summary_${TAG}_$PREC.json $PREC 140 > gpurun_out/pmc_per_op_${TAG}_$PREC.txt 2>&1; tail -12 gpurun_out/pmc_per_op_${TAG}_$PREC.txt
rm -rf gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write
