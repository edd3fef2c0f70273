#!/bin/bash
# round 5, GPU call 20: k_stage2 with compact distinct-token tables (one table access per token) and the tokens' byte lengths in LDS — parity subset, then A/B on configs 3 and 4
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c20; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "synthetic_parity or long_documents or long_queries or ten_docs or unicode or scripts or long_tokens or wordmatcher_lists or reference_kats" > $O/parity.log 2>&1; echo "parity rc=$?" >> $O/parity.log )
tail -5 $O/parity.log
run() { # name lib pool config steps
  INFX_LIB=$2 INFX_S2_POOL=$3 timeout 300 python bench.py --config $4 --steps $5 --warmup 3 --no-cpu-baseline --sessions 1 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'ms/step %.2f' % d['ms_per_step'], 'k_stage2 %.3f' % d['roofline']['other_kernels_ms']['k_stage2'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
M=$R/infidex_amd/libinfidex_hip.so; V=$R/infidex_amd/libinfidex_hip_culen0.so
run cfg3_lds_p3072 $M 3072 3 30; run cfg3_lds_p2560 $M 2560 3 30; run cfg3_lds_p2048 $M 2048 3 30; run cfg3_compact_p3072 $V 3072 3 30
run cfg4_lds_p3072 $M 3072 4 8; run cfg4_lds_p2048 $M 2048 4 8; run cfg4_compact_p3072 $V 3072 4 8
