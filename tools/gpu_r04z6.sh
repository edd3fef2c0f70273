#!/bin/bash
mkdir -p gpurun_out/r04z6
timeout 40 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04z6/bench.json 2> gpurun_out/r04z6/bench.err; echo "rc=$?"
python -c "
import json
j=json.loads(open('gpurun_out/r04z6/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],2), round(j['p50_batch_latency_ms'],1), round(j['p95_batch_latency_ms'],1), j['config'])"
