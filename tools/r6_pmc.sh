#!/bin/bash
# round 6: PMC passes for k_accumulate (each its own run; --pmc never together with trace flags).  usage: tools/r6_pmc.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
for kv in "$@"; do export $kv; done
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --sessions 1 --long-steps 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p --output-format csv -- $B > $O/pmc_sq.json 2> $O/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p --output-format csv -- $B > $O/pmc_sq2.json 2> $O/pmc_sq2.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $O/pmc_sq3 -o p --output-format csv -- $B > $O/pmc_sq3.json 2> $O/pmc_sq3.err
for d in pmc_fetch pmc_sq pmc_sq2 pmc_sq3; do
  python $R/tools/pmc_summary.py $O/$d "k_accumulate" > $O/$d.txt 2>&1
  rm -rf $O/$d
done
cat $O/pmc_fetch.txt $O/pmc_sq.txt $O/pmc_sq2.txt $O/pmc_sq3.txt; head -8 $O/kernel_stats.csv | cut -c1-60,150-260
