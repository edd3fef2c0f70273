#!/bin/bash
# round 5, GPU call 18: LDS text pool of k_stage2's fast launch on config 3 (documents 2.6 x longer than config 4's, for which 3072 units were tuned)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c18; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for pool in 3072 6144 8192 12288 16384; do
  INFX_S2_POOL=$pool timeout 300 python bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --sessions 1 > $O/cfg3_pool$pool.json 2> $O/cfg3_pool$pool.err
  python - $O/cfg3_pool$pool.json $pool <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('pool', sys.argv[2], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], 'k_stage2 %.3f' % d['roofline']['other_kernels_ms']['k_stage2'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
