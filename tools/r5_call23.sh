#!/bin/bash
# round 5, GPU call 23: LD1 placement threshold 2.5 x -> 1.0 x threads: config 3 must pick the device on its own, config 4 the host as before (short run)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c23; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline > $O/cfg3.json 2> $O/cfg3.err
timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/cfg4.json 2> $O/cfg4.err
for f in cfg3 cfg4; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], d.get('planning_lookups'), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
