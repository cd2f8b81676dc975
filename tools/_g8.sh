mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-second-mode --repeats 1 --steps 6 --warmup 2 $EXTRA > gpurun_out/bench_$tag.log 2>&1; tail -1 gpurun_out/bench_$tag.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'])
ops=json.load(open('gpurun_out/ops_$tag.json'))
print('   ', ' '.join('%s=%.4f' % (o['name'].split('_')[1]+'_'+o['name'].split('_')[2], o['ms_per_launch']) for o in ops if 'par4' in o['name'] or o['name'].startswith('conv3x3')))
"; }
EXTRA="" run m0 SBBSEG_XR_MASK=0 SBBSEG_BENCH_OPS=gpurun_out/ops_m0.json
EXTRA="" run m15e SBBSEG_XR_MASK=15 SBBSEG_BENCH_OPS=gpurun_out/ops_m15e.json
EXTRA="--conv-variant 262144" run m15l SBBSEG_XR_MASK=15 SBBSEG_BENCH_OPS=gpurun_out/ops_m15l.json
