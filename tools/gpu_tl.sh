#!/bin/bash
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; env "$@" INFX_BENCH_TIMELINE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 >/tmp/out.json | grep timeline | awk '{ if ($13+0 > mw) mw=$13+0; if ($11+0 > mp) mp=$11+0; if ($9+0 > end) end=$9+0 } END { printf "end %.1f ms  max wait %.1f  max plan %.1f\n", end, mw, mp }'; python -c "
import json; d=json.load(open('/tmp/out.json')); print(round(d['value']), round(d['p50_batch_latency_ms'],1), round(d['p95_batch_latency_ms'],1))"; }
EXTRA="" run default A=1
EXTRA="" run poll INFX_SYNC_POLL=1
EXTRA="" run poll2 INFX_SYNC_POLL=1
EXTRA="" run poll3 INFX_SYNC_POLL=1
EXTRA="" run default2 A=1
