#!/bin/bash
# round 5, GPU call 16: closing validation of the final tree — smoke (with a long query), kernel trace of the final build, full GPU suite, the driver's bench form
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r5c16; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt )
tail -3 $O/smoke.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1 > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
cd $R
head -12 $O/kernel_stats.csv | cut -c1-150
( timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
tail -4 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
python - $O/bench_20.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('bench_20', round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), d['cpu_baseline'].get('identical_topk_sets'), d['roofline']['frac'], d['roofline']['other_kernels_ms'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
