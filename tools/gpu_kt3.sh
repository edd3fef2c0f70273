#!/bin/bash
for v in "full:INFX_ACC_SR=1" "nolist:INFX_ACC_SR=1 INFX_ACC_SKIP=256" "nohit:INFX_ACC_SR=1 INFX_ACC_SKIP=512"; do
  name=${v%%:*}; envs=${v#*:}
  OUT=kt_sr_$name bash $GRAFT_REPO_ROOT/tools/gpu_kt.sh $envs > /dev/null 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/kt_sr_$name/kernel_stats.csv $name <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    n=r['Name']
    if 'k_accumulate' in n: print(sys.argv[2], n.split('(')[0][:40], r['Calls'], 'avg %.3f ms' % (float(r['AverageNs'])/1e6), 'max %.3f' % (int(r['MaxNs'])/1e6))
PY
done
