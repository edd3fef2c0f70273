#!/bin/bash
# round 6: kernel-trace A/B of library variants.  usage: tools/r6_ab.sh <tag> <lib suffix or "base">... [-- ENV=.. ...]   (bench: 12 steps, no CPU baseline)
R=$GRAFT_REPO_ROOT; TAG=$1; shift
for v in "$@"; do
  lib=$R/infidex_amd/libinfidex_hip.so; [ "$v" != base ] && lib=$R/infidex_amd/libinfidex_hip_$v.so
  echo "== $v"
  INFX_LIB=$lib bash $R/tools/r6_kt.sh ${TAG}_$v -- --steps 12 --warmup 3 --long-steps 0 --no-cpu-baseline $BARGS | grep "${KGREP:-k_accumulate}"
  python - $R/gpurun_out/${TAG}_$v/kt.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('  bench', round(d['value']), 'acc_ms %.3f' % d['roofline']['avg_launch_ms'])
except Exception as e: print('  ERR', e)
PY
done
