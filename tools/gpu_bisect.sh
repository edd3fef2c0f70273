#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT/.bisect/r03 && timeout 600 python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/c2_r03.json 2> $O/c2_r03.err
cd $GRAFT_REPO_ROOT && timeout 600 python bench.py --config 2 --steps 40 --warmup 5 --no-cpu-baseline > $O/c2_now.json 2> $O/c2_now.err
cd $GRAFT_REPO_ROOT/.bisect/r03 && timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/c4_r03.json 2> $O/c4_r03.err
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r04j/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
