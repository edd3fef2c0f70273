#!/bin/bash
# round 6: rocprofv3 kernel trace of one bench run.  usage: tools/r6_kt.sh <tag> [env assignments...] -- <bench args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
while [ "$1" != "--" ] && [ -n "$1" ]; do export $1; shift; done; shift
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/bench.py "$@" > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
python - $O/kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), r['Percentage'].rjust(7), 'max %.1f' % (float(r['MaxNs'])/1e3))
PY
