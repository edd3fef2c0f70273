#!/bin/bash
# k_accumulate profiling session on the GPU box: kernel trace, PMC passes (each its own run) and ablations.  Output under gpurun_out/prof_acc/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_acc; mkdir -p $O
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sessions 1"
export INFX_EXACT=${INFX_EXACT:-0}
$B > $O/base.json 2> $O/base.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc1 -o p --output-format csv -- $B > $O/pmc1.json 2> $O/pmc1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/pmc2 -o p --output-format csv -- $B > $O/pmc2.json 2> $O/pmc2.err
for k in 1 2 4; do INFX_ACC_SKIP=$k $B > $O/skip$k.json 2> $O/skip$k.err; done
for st in 1 8; do INFX_ACC_STRIPE=$st $B > $O/stripe$st.json 2> $O/stripe$st.err; done
python $R/tools/pmc_summary.py $O/pmc1 k_accumulate > $O/pmc1.txt 2>&1
python $R/tools/pmc_summary.py $O/pmc2 k_accumulate > $O/pmc2.txt 2>&1
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
for f in base skip1 skip2 skip4 stripe1 stripe8; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["other_kernels_ms"])
except Exception as e: print("$f", "ERR", e)
PY
done > $O/summary.txt
rm -rf $O/kt/*/*.db $O/pmc1/*/*.db $O/pmc2/*/*.db 2>/dev/null
cat $O/summary.txt $O/pmc1.txt $O/pmc2.txt; head -12 $O/kernel_stats.csv
