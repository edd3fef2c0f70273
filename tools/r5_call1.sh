#!/bin/bash
# round 5, GPU call 1: staged non-BMP test, full GPU suite, replay profile (register heap vs LDS heap), bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c1; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( INFX_RUN_STAGED=1 timeout 600 python -m pytest tests/test_gpu_staged.py -m gpu -x -q > $O/staged.log 2>&1; echo "staged rc=$?" >> $O/staged.log )
( timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
INFX_EXACT_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --sessions 1 --no-cpu-baseline > $O/prof_reg.json 2> $O/prof_reg.err
INFX_EXACT_PROF=1 INFX_EX_HEAP_LDS=1 timeout 300 python bench.py --steps 3 --warmup 1 --sessions 1 --no-cpu-baseline > $O/prof_lds.json 2> $O/prof_lds.err
timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
tail -3 $O/staged.log; tail -3 $O/gputest.log
grep -h "infx\]" $O/prof_reg.err | tail -40
echo ---- LDS heap; grep -h "heap\.\|infx\] exact" $O/prof_lds.err | tail -8
for f in bench_s1 bench_20; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), d['roofline'].get('other_kernels_ms'), [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d.get('roofline_by_kernel',[])])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
