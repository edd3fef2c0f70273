#!/bin/bash
# ablation timings of k_accumulate2 (INFX_ACC_DBG) — profiling only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl; mkdir -p $O
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sessions 1"
export INFX_EXACT=0
for d in 0 1 2 3 16 32 48 64 19 35 18 34; do INFX_ACC_DBG=$d $B > $O/d$d.json 2> $O/d$d.err; done
for d in 0 1 2 3 16 32 48 64 19 35 18 34; do python - <<PY
import json
try:
    d=json.loads(open("$O/d$d.json").read().strip().splitlines()[-1]); print("dbg $d", "acc_ms", round(d["roofline"]["avg_launch_ms"],3))
except Exception as e: print("dbg $d", "ERR", e)
PY
done
