#!/bin/bash
# round 5, GPU call 22: smoke on the final binary; config 3 with the LD1 expansion pinned to the device (k_ld1 throughput next to the host walk)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c22; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt ); tail -2 $O/smoke.txt
for v in "1 4" "0 4" "1 1" "0 1"; do
  set -- $v
  INFX_DEVICE_LOOKUPS=$1 timeout 200 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline --sessions $2 > $O/cfg3_devld1_$1_s$2.json 2> $O/cfg3_devld1_$1_s$2.err
  python - $O/cfg3_devld1_$1_s$2.json $1 $2 <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('device_ld1', sys.argv[2], 'sessions', sys.argv[3], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d.get('planning_lookups'), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
