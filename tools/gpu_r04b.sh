#!/bin/bash
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" >> $O/summary.txt; }
EXTRA="" run cfg4_dev A=1
EXTRA="" run cfg4_noprio INFX_PLAN_PRIORITY=0
EXTRA="--sessions 4" run cfg4_s4 A=1
EXTRA="--sessions 3" run cfg4_s3 A=1
EXTRA="--config 3" run cfg3_dev A=1
EXTRA="--config 3" run cfg3_host INFX_HOST_LOOKUPS=1
timeout 900 python -m pytest tests/test_gpu_sharded_ranks.py tests/test_gpu_lookups.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04b/*.json')):
    try:
        d=json.load(open(f)); s=d['stage_ms_per_step']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']),
              {k: round(v,2) for k,v in s.items() if k.startswith('plan') or k in ('prep2_ms','post_ms','stage2_ms')})
    except Exception as e: print(f, 'ERR', e)
PY
