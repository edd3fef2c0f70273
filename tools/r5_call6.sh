#!/bin/bash
# round 5, GPU call 6: k_select longest-first order A/B, segment-sourced engine test, Unicode scripts test, config-3 kernel trace
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c6; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_infs.py tests/test_gpu_scale.py "tests/test_gpu_parity.py::test_scripts_beyond_latin1_full_parity" "tests/test_gpu_parity.py::test_ordinal_ignore_case_aliases" "tests/test_gpu_parity.py::test_synthetic_parity" -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
INFX_SELECT_LPT=1 timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/lpt1.json 2> $O/lpt1.err
INFX_SELECT_LPT=0 timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/lpt0.json 2> $O/lpt0.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt3 -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --sessions 1 > $GRAFT_REPO_ROOT/$O/kt3.json 2> $GRAFT_REPO_ROOT/$O/kt3.err
cd $GRAFT_REPO_ROOT
find $O/kt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg3.csv; rm -rf $O/kt3
tail -6 $O/gputest.log
head -14 $O/kernel_stats_cfg3.csv | cut -c1-60,200-
for f in lpt1 lpt0 bench_20 kt3; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc %.3f' % d['roofline']['avg_launch_ms'], [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d.get('roofline_by_kernel',[])], {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
