#!/bin/bash
# kernel trace of a short single-session bench run with the environment given as arguments; prints the top kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-kt}; mkdir -p $O
env "$@" timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1 > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
head -12 $O/kernel_stats.csv | cut -c1-60,200-330
