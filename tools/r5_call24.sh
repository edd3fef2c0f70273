#!/bin/bash
# round 5, GPU call 24: bench lines of configs 2 and 5 on the final binary
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c24; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python bench.py --config 2 --steps 60 --warmup 5 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err
timeout 100 python bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $O/cfg5.json 2> $O/cfg5.err
for f in cfg2 cfg5; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
