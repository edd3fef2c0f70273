#!/bin/bash
# round 6 GPU call driver: tools/r6_call.sh <tag> <steps...>   (steps: test | testfile:<path> | bench:<name>:<env assignments,comma separated>:<bench args>)
R=$GRAFT_REPO_ROOT; cd $R; TAG=$1; shift; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  case $step in
    test) ( timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log ); tail -5 $O/gputest.log ;;
    testfile:*) f=${step#testfile:}; n=$(echo $f | tr '/:' '__'); ( timeout 1500 python -m pytest $f -m gpu -q -x > $O/test_$n.log 2>&1; echo "rc=$?" >> $O/test_$n.log ); tail -15 $O/test_$n.log ;;
    bench:*) IFS=: read -r _ name envs args <<< "$step"
      ( for kv in ${envs//,/ }; do export $kv; done; timeout 900 python bench.py $args > $O/$name.json 2> $O/$name.err )
      python - $O/$name.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc_ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), (d.get('cpu_baseline') or {}).get('identical_topk_sets'), r.get('other_kernels_ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
      ;;
  esac
done
