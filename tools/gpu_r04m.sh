#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_lookups.py tests/test_gpu_fullsize.py tests/test_gpu_sharded_ranks.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "eviction or kat or fuzzy" 2>&1 | tail -2
one() { env $2 python bench.py --steps $1 --warmup 5 --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms_per_step']; print('$2 $3 steps $1:', round(d['value']), round(d['ms_per_step'],2), 'p50', round(d['p50_batch_latency_ms'],1), 'p95', round(d['p95_batch_latency_ms'],1), {k: round(v,1) for k,v in s.items() if k.startswith('plan')}, d['planning_lookups'])"; }
one 60 A=1 "--config 3"
one 60 INFX_LD1_FUSED=0 "--config 3"
one 60 "INFX_DEVICE_LOOKUPS=1" ""
one 60 "INFX_DEVICE_LOOKUPS=1 INFX_LD1_FUSED=0" ""
