#!/bin/bash
# round 5, GPU call 12: plan exchange — full GPU suite, two gloo ranks on one GPU at full size with the exchange on and off, the driver's bench form
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c12; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
tail -4 $O/gputest.log
for px in 1 0; do
  INFX_PLAN_EXCHANGE=$px INFX_DIST_BACKEND=gloo MASTER_PORT=2964$px timeout 600 python bench.py --gpus 2 --steps 12 --warmup 3 --no-cpu-baseline > $O/two_ranks_px$px.json 2> $O/two_ranks_px$px.err
  python - $O/two_ranks_px$px.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], [{k: round(v,2) for k,v in r.items() if k.startswith('plan')} for r in d['stage_ms_per_step_per_rank']], [c.get('plan_exchange') for c in d['collectives_per_rank']])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
python - $O/bench_20.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('bench_20', round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), d.get('parity'), d['roofline']['frac'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
