#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_native_driver_device_buffers.py tests/test_gpu_sharded_ranks.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
INFX_FORCE_SHARDED=1 timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/sharded_w1.json 2> $O/sharded_w1.err; echo "sharded rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -15 $O/pytest.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c/sharded_w1.json'))
print(round(d['value']), d['ms_per_step'], d['p50_batch_latency_ms'], d.get('collectives_per_rank'), d['stage_ms_per_step'])
PY
