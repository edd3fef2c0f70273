#!/bin/bash
# k_accumulate3 vs k_accumulate: the A/B parity test, then the 10 M-doc bench with each (short runs, no CPU baseline).  Output: gpurun_out/acc3/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/acc3; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "accumulate_designs" > $O/ab.log 2>&1; tail -3 $O/ab.log
INFX_ACC_V3=1 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 4 > $O/bench_v3.json 2> $O/bench_v3.err
INFX_ACC_V3=0 timeout 600 python bench.py --no-cpu-baseline --steps 40 --warmup 4 > $O/bench_v1.json 2> $O/bench_v1.err
for f in v3 v1; do python - $O/bench_$f.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["value"]), "q/s  acc_ms", round(d["roofline"]["avg_launch_ms"],3), "frac", round(d["roofline"]["frac"],3))
PY
done
