#!/bin/bash
# ablations of k_accumulate3 / k_accumulate (INFX_ACC_DBG / INFX_ACC_SKIP: results meaningless, timing only).  Output: gpurun_out/acc3/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/acc3; mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > $O/abl_$tag.json 2> $O/abl_$tag.err; python - $O/abl_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "acc_ms", round(d["roofline"]["avg_launch_ms"],3))
PY
}
run v3_full INFX_ACC_V3=1
run v3_nolist INFX_ACC_V3=1 INFX_ACC_DBG=16
run v3_noterm INFX_ACC_V3=1 INFX_ACC_DBG=32
run v3_neither INFX_ACC_V3=1 INFX_ACC_DBG=48
run v1_full INFX_ACC_V3=0
run v1_nolist INFX_ACC_V3=0 INFX_ACC_SKIP=1
