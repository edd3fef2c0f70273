#!/bin/bash
export TMPDIR=/tmp
one() { env $2 python bench.py --steps $1 --warmup 5 --no-cpu-baseline $3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$2 $3 steps $1:', round(d['value']), round(d['ms_per_step'],2), 'p50', round(d['p50_batch_latency_ms'],1), 'p95', round(d['p95_batch_latency_ms'],1))"; }
for s in 3 4 5; do one 100 A=1 "--sessions $s"; one 100 INFX_TURNSTILE=0 "--sessions $s"; done
one 20 INFX_TURNSTILE=0 "--sessions 4"; one 20 INFX_TURNSTILE=0 "--sessions 5"
