#!/bin/bash
# round 5, GPU call 21: closing validation of the final tree (compact distinct-token tables in k_stage2) — full GPU suite, bench lines of config 4 (driver form) and config 3
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c21; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
tail -6 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
timeout 600 python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
for f in bench_20 bench_cfg3; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), (d.get('cpu_baseline') or {}).get('identical_topk_sets'), d['roofline']['other_kernels_ms'], {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
