#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "accumulate_designs" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $EXTRA > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" >> $O/summary.txt; }
EXTRA="" run c4_base A=1
EXTRA="" run c4_sr INFX_ACC_SR=1
EXTRA="" run c4_noprio INFX_PLAN_PRIORITY=0
EXTRA="" run c4_q16 GPU_MAX_HW_QUEUES=16
EXTRA="--config 2" run c2_base A=1
EXTRA="--config 2" run c2_noprio INFX_PLAN_PRIORITY=0
EXTRA="--config 2" run c2_q16 GPU_MAX_HW_QUEUES=16
EXTRA="--config 2" run c2_host INFX_HOST_LOOKUPS=1
cat $O/summary.txt; tail -4 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04i/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
