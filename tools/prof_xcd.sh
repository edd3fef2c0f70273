#!/bin/bash
# XCD-aware block map on / off for k_accumulate and k_accumulate3 (bench, short runs) + FETCH_SIZE of k_accumulate with the map on.  Output: gpurun_out/xcd/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/xcd; rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 4 > $O/$tag.json 2> $O/$tag.err; python - $O/$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "q/s acc_ms", round(d["roofline"]["avg_launch_ms"],3))
PY
}
run v1_xcd INFX_ACC_V3=0
run v1_old INFX_ACC_V3=0 INFX_ACC_SKIP=32
run v3_xcd INFX_ACC_V3=1
run v3_old INFX_ACC_V3=1 INFX_ACC_DBG=128
cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --sessions 1"
INFX_ACC_V3=0 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/f -o p --output-format csv -- $B > $O/f.json 2> $O/f.err
python $R/tools/pmc_summary.py $O/f "k_accumulate<8192, 2>"; rm -rf $O/f
