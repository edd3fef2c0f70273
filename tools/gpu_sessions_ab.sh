#!/bin/bash
# sessions in flight for the driver's 20-step form, same box, one after the other
mkdir -p gpurun_out/sessions_ab
for s in 4 5 3; do
  timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sessions $s > gpurun_out/sessions_ab/bench_s$s.json 2> gpurun_out/sessions_ab/bench_s$s.err
  python - $s <<'PY'
import json, sys
s = sys.argv[1]
try:
    j = json.loads(open(f"gpurun_out/sessions_ab/bench_s{s}.json").read().strip().splitlines()[-1])
    print("sessions", s, round(j["value"]), "q/s", round(j["ms_per_step"], 2), "p50", round(j["p50_batch_latency_ms"], 1), "p95", round(j["p95_batch_latency_ms"], 1))
except Exception as e:
    print("sessions", s, "failed", e)
PY
done
