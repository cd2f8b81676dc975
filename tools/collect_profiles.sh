# One GPU-box session: full GPU test suite, smoke, the default bench line, rocprofv3 kernel stats + per-op trace table, PMC passes.
# usage: bash tools/collect_profiles.sh <tag>   (results under gpurun_out/; copy what should be judged into profiles/)
TAG=${1:-r02}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
SBBSEG_BENCH_OPS=gpurun_out/ops_$TAG.json timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log > gpurun_out/bench_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print("BENCH", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("frac_issued"), "cpu", d["cpu_baseline"])
PY
REPO=$(pwd)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-second-mode > $REPO/gpurun_out/rocprof_$TAG.log 2>&1)
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" gpurun_out/kernel_stats_$TAG.csv && head -8 "$F"
T=$(find gpurun_out/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$T" ] && python tools/trace_per_op.py "$T" gpurun_out/ops_$TAG.json > gpurun_out/kernel_trace_per_op_$TAG.md 2>&1; tail -5 gpurun_out/kernel_trace_per_op_$TAG.md
rm -rf gpurun_out/prof_$TAG
BENCH_ARGS="--no-second-mode" bash tools/pmc_run.sh pmc_$TAG > gpurun_out/pmc_run_$TAG.log 2>&1
python tools/pmc_report.py pmc_$TAG gpurun_out/ops_$TAG.json gpurun_out/pmc_summary_$TAG.json f16 140 > gpurun_out/pmc_per_op_$TAG.txt 2>&1; tail -12 gpurun_out/pmc_per_op_$TAG.txt
rm -rf gpurun_out/pmc_${TAG}_sq gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write
