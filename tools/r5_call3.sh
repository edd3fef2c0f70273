#!/bin/bash
# round 5, GPU call 3: two-rank bench (stderr kept), parity suites after the Unicode / theta changes, replay profile, bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( INFX_DIST_BACKEND=gloo MASTER_PORT=29641 INFX_THREADS=4 timeout 600 python bench.py --gpus 2 --docs 140000 --steps 2 --warmup 1 --batch 200 --no-cpu-baseline > $O/bench_w2.json 2> $O/bench_w2.err; echo "rc=$?" >> $O/bench_w2.err )
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_lookups.py tests/test_gpu_schools.py tests/test_golden.py -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
INFX_EXACT_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --sessions 1 --no-cpu-baseline > $O/prof_reg.json 2> $O/prof_reg.err
timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/bench_s1.json 2> $O/bench_s1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sessions 3 > $O/bench_20_s3.json 2> $O/bench_20_s3.err
grep -v "^\s*$" $O/bench_w2.err | grep -v Warning | tail -30
tail -12 $O/gputest.log
grep -h "replayed queries of 1000" -A 22 $O/prof_reg.err | head -24
for f in prof_reg bench_s1 bench_20 bench_20_s3; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc %.2f' % d['roofline']['avg_launch_ms'], [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d.get('roofline_by_kernel',[])])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
