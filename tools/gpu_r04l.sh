#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -k "replay or oracle_sample or deleted or exact" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" >> $O/summary.txt; }
EXTRA="--steps 60 --warmup 5" run aux A=1
EXTRA="--steps 60 --warmup 5" run noaux INFX_REPLAY_AUX=0
EXTRA="--steps 60 --warmup 5" run aux2 A=1
EXTRA="--steps 60 --warmup 5" run noaux2 INFX_REPLAY_AUX=0
EXTRA="--steps 20 --warmup 5 --sessions 4" run s4_20 A=1
EXTRA="--steps 20 --warmup 5 --sessions 5" run s5_20 A=1
EXTRA="--steps 20 --warmup 5 --sessions 6" run s6_20 A=1
EXTRA="--steps 20 --warmup 5 --sessions 8" run s8_20 A=1
cat $O/summary.txt; tail -3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04l/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; s=d['stage_ms_per_step']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()}, 'plan %.1f' % s['plan_ms'], d.get('planning_lookups'))
    except Exception as e: print(f, 'ERR', e)
PY
