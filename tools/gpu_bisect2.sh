#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
cd $GRAFT_REPO_ROOT/.bisect/r03 && timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/c4_r03_$i.json 2> $O/c4_r03_$i.err
cd $GRAFT_REPO_ROOT && timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/c4_now_$i.json 2> $O/c4_now_$i.err
cd $GRAFT_REPO_ROOT && INFX_HOST_LOOKUPS=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/c4_nowhost_$i.json 2> $O/c4_nowhost_$i.err
cd $GRAFT_REPO_ROOT && INFX_PLAN_PRIORITY=0 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/c4_nowNoprio_$i.json 2> $O/c4_nowNoprio_$i.err
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04k/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; s=d['stage_ms_per_step']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()}, 'plan %.1f prep2 %.1f wait %.1f post %.2f' % (s['plan_ms'], s['prep2_ms'], s['stage2_ms'], s['post_ms']))
    except Exception as e: print(f, 'ERR', e)
PY
