#!/bin/bash
# round 5, GPU call 19: k_stage2 on config 3 — smaller text pools and the register budget of the fast launch
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c19; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in "0 6" "1024 6" "2048 6" "3072 6" "3072 8" "3072 4" "2048 8" "1024 8"; do
  set -- $v
  INFX_S2_POOL=$1 INFX_S2_WAVES=$2 timeout 300 python bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --sessions 1 > $O/cfg3_p$1_w$2.json 2> $O/cfg3_p$1_w$2.err
  python - $O/cfg3_p$1_w$2.json $1 $2 <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('pool', sys.argv[2], 'waves', sys.argv[3], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], 'k_stage2 %.3f' % d['roofline']['other_kernels_ms']['k_stage2'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
