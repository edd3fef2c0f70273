#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_native_driver_device_buffers.py tests/test_gpu_sharded_ranks.py -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
INFX_FORCE_SHARDED=1 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/sharded_ring.json 2> $O/sharded_ring.err; echo "ring rc=$?" >> $O/summary.txt
INFX_FORCE_SHARDED=1 INFX_COLL_ORDER=0 timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/sharded_noring.json 2> $O/sharded_noring.err; echo "noring rc=$?" >> $O/summary.txt
cat $O/summary.txt; grep -v "^$" $O/pytest.log | tail -12
python - <<'PY'
import json
for n in ('sharded_ring','sharded_noring'):
    d=json.load(open(f'gpurun_out/r04d/{n}.json'))
    print(n, round(d['value']), round(d['ms_per_step'],2), round(d['p50_batch_latency_ms'],1), d.get('collectives_per_rank'))
PY
