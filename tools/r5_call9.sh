#!/bin/bash
# round 5, GPU call 9: k_ex_theta insert position without eight ballots: parity suites + bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c9; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_sharded_ranks.py "tests/test_gpu_parity.py::test_synthetic_parity" "tests/test_gpu_parity.py::test_sharded_equals_the_oracle" -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/s1.json 2> $O/s1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
tail -6 $O/gputest.log
for f in s1 bench_20; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc %.3f' % d['roofline']['avg_launch_ms'], [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d.get('roofline_by_kernel',[])])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
