#!/bin/bash
mkdir -p gpurun_out/r04z4
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04z4/bench_20steps.json 2> gpurun_out/r04z4/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r04z4/bench_20steps.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("bench", round(j["value"]), "q/s", round(j["ms_per_step"], 2), "frac", round(r["frac"], 3), "of copy", round(r["frac_of_measured_copy_rate"], 3), "traffic", r["traffic"])
PY
