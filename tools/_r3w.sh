mkdir -p gpurun_out
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r03_final.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/bench_r03_final.log > gpurun_out/bench_r03_final.json
python -c "
import json; d=json.load(open('gpurun_out/bench_r03_final.json')); print(d['value'], d['roofline']['traffic'], d['roofline'].get('traffic_static'), d['modes']['f16']['value'] if 'value' in d['modes']['f16'] else d['modes']['f16'])" 2>&1 | cut -c1-600
