#!/bin/bash
# kernel trace of a short bench run (all kernels); output: gpurun_out/prof_kt/kernel_stats.csv + bench json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_kt; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sessions 1 $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/kt_kernel_trace.csv")))
import collections
d=collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0][:40]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    big=[x for x in v if x>0.05]
    print(f"{k:42s} n={len(v):4d} total={sum(v):9.3f} ms  max={max(v):8.3f}  mean(>50us)={sum(big)/max(1,len(big)):8.3f} n_big={len(big)}")
PY
tail -c 1500 $O/kt.json
rm -f $O/kt/*kernel_trace.csv
