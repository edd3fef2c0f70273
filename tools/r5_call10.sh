#!/bin/bash
# round 5, GPU call 10: do the planning streams queue behind other sessions' kernels because 4 sessions x 3 streams share 8 hardware queues?  config 3 and config 4 with 8 / 16 / 24 queues
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c10; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for hq in 8 16 24; do
  GPU_MAX_HW_QUEUES=$hq timeout 300 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg3_q$hq.json 2> $O/cfg3_q$hq.err
  GPU_MAX_HW_QUEUES=$hq timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg4_q$hq.json 2> $O/cfg4_q$hq.err
done
for f in cfg3_q8 cfg3_q16 cfg3_q24 cfg4_q8 cfg4_q16 cfg4_q24; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
