#!/bin/bash
# A/B of k_accumulate variants selected by INFX_ACC_SKIP bits: bench lines side by side.  usage: gpu_ab.sh OUTDIR "name:ENV=VAL" ...
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs timeout 600 python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline ${EXTRA} > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" >> $O/summary.txt
done
cat $O/summary.txt
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
