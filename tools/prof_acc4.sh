#!/bin/bash
# k_accumulate4 vs k_accumulate: the five-way A/B parity test, then the 10 M-doc bench with each (short runs, no CPU baseline).  Output: gpurun_out/acc4/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/acc4; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "accumulate_designs" > $O/ab.log 2>&1; tail -3 $O/ab.log
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 40 --warmup 4 > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "q/s acc_ms", round(d["roofline"]["avg_launch_ms"],3), "frac", round(d["roofline"]["frac"],3))
PY
}
run v4_sup2 INFX_ACC_V4=1 INFX_ACC_SUP=2
run v4_sup4 INFX_ACC_V4=1 INFX_ACC_SUP=4
run v1 INFX_ACC_V4=0
