#!/bin/bash
# bench lines of the other BASELINE configs (with the CPU baseline / parity block) and of the sharded path on one GPU.  Output: gpurun_out/prof_r03/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; mkdir -p $O
python $R/bench.py --config 2 --steps 50 --warmup 3 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python $R/bench.py --config 3 --steps 50 --warmup 3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python $R/bench.py --config 5 --steps 60 --warmup 4 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
INFX_FORCE_SHARDED=1 python $R/bench.py --no-cpu-baseline --steps 64 --warmup 4 > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
for f in cfg2 cfg3 cfg5 sharded_w1 20steps; do tail -c 700 $O/bench_$f.json; echo; done
