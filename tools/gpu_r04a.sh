#!/bin/bash
# round 4, first GPU pass: new lookup tests first, then the whole GPU suite, then bench lines (device vs host lookups, config 3)
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lookups.py -x -q > $O/pytest_lookups.log 2>&1; echo "lookups rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_lookups.py > $O/pytest_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_dev.json 2> $O/bench_cfg4_dev.err; echo "bench4 rc=$?" >> $O/summary.txt
INFX_HOST_LOOKUPS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_host.json 2> $O/bench_cfg4_host.err; echo "bench4host rc=$?" >> $O/summary.txt
timeout 600 python bench.py --config 3 --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_cfg3_dev.json 2> $O/bench_cfg3_dev.err; echo "bench3 rc=$?" >> $O/summary.txt
tail -3 $O/pytest_lookups.log; tail -3 $O/pytest_all.log; cat $O/summary.txt
