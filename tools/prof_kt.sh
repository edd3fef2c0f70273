#!/bin/bash
# kernel trace of a short single-session bench run -> gpurun_out/prof_kt/kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_kt; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/kt
cut -c1-60 $O/kernel_stats.csv | head -5; python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    print(r["Name"][:40].ljust(40), r["Calls"].rjust(5), ("%.3f" % (float(r["AverageNs"])/1e6)).rjust(9), ("%.3f" % (float(r["MaxNs"])/1e6)).rjust(9), r["Percentage"].rjust(7))
PY
