#!/bin/bash
# round 5, GPU call 15: long queries + quirk Q18 — the tests that failed in call 13 first, then the full suite, then the driver's bench form
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c15; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest "tests/test_gpu_parity.py::test_long_queries_take_the_long_query_launches" "tests/test_gpu_parity.py::test_long_documents_take_the_retry_launches" "tests/test_gpu_parity.py::test_sharded_equals_the_oracle" "tests/test_gpu_scale.py::test_host_phase_implementation_equals_device_pipeline" "tests/test_infdx2.py::test_loaded_index_searches_like_the_indexed_one" "tests/test_gpu_parity.py::test_query_with_more_wordmatcher_lists_than_the_device_limit" -m gpu -q > $O/longq.log 2>&1; echo "longq rc=$?" >> $O/longq.log )
tail -30 $O/longq.log
( timeout 1200 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
tail -6 $O/gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
python - $O/bench_20.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print('bench_20', round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), d['roofline']['other_kernels_ms'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
