#!/bin/bash
# round-4 closing validation of the final build: the whole GPU suite, the smoke test, the driver's bench form (CPU baseline skipped: 133 s of oracle index build)
mkdir -p gpurun_out/closing
export INFX_COMM_TIMEOUT_S=60
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/closing/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/closing/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/closing/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/closing/smoke.log
timeout 110 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/closing/bench_20steps.json 2> gpurun_out/closing/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/closing/bench_20steps.json").read().strip().splitlines()[-1])
    print("bench", round(j["value"]), "q/s", j["ms_per_step"], "p50", j["p50_batch_latency_ms"], "p95", j["p95_batch_latency_ms"], j.get("parity"), (j.get("roofline") or {}).get("traffic"))
except Exception as e:
    print("bench parse failed", e)
PY
