#!/bin/bash
# Everything the end of a round needs from ONE gpurun call (about 8 GPU-minutes):
#   gpurun --timeout 3600 -- 'bash tools/round_end.sh r04'
# full `pytest -m gpu`, smoke(), the default bench line, per-op rocprofv3 trace + three PMC passes for both arithmetic modes, the N > 1
# code path of bench.py on one GPU (gloo-staged exchange), and -- last, so that it can use the PMC summaries just written -- the bench
# line again.  Outputs land in gpurun_out/; copy the ones to be judged into profiles/ (see profiles/README.md) and run the last step
# once more from the committed tree if `roofline.traffic` is to come from the committed summaries.
tag=${1:-r06}
bash tools/collect_profiles.sh $tag f16x3
bash tools/collect_profiles.sh $tag f16 skip-tests
SBBSEG_BENCH_BACKEND=gloo timeout -k 5 900 python bench.py --gpus 2 --batch-pages 8 --steps 2 --warmup 1 --repeats 1 > gpurun_out/bench_${tag}_gloo2.log 2>&1; echo "gloo2 rc=$?"
tail -1 gpurun_out/bench_${tag}_gloo2.log > gpurun_out/bench_${tag}_gloo2.json
# the PMC summaries just written become the ones bench.py reads `roofline.traffic` from (profiles/<tag>_{x3,f16}_pmc_summary.json: commit
# the copies that come back under gpurun_out/ under exactly these names)
cp gpurun_out/pmc_summary_${tag}_f16x3.json profiles/${tag}_x3_pmc_summary.json 2>/dev/null
cp gpurun_out/pmc_summary_${tag}_f16.json profiles/${tag}_f16_pmc_summary.json 2>/dev/null
timeout -k 5 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_${tag}_final.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/bench_${tag}_final.log > gpurun_out/bench_${tag}_final.json
python -c "
import json; d=json.load(open('gpurun_out/bench_${tag}_final.json')); print(d['value'], d['roofline']['traffic'], d['roofline'].get('modes'))"
