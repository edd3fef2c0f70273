#!/bin/bash
# Where do k_accumulate's wave cycles go?  Three PMC passes (own runs, no trace flags).  Output: gpurun_out/stall/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stall; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $O/a -o p --output-format csv -- $B > $O/a.json 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL -d $O/b -o p --output-format csv -- $B > $O/b.json 2> $O/b.err
rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_LEVEL_WAVES -d $O/c -o p --output-format csv -- $B > $O/c.json 2> $O/c.err
for d in a b c; do python $R/tools/pmc_summary.py $O/$d "k_accumulate<8192, 2>" > $O/$d.txt 2>&1; rm -rf $O/$d; done
cat $O/a.txt $O/b.txt $O/c.txt; tail -3 $O/a.err
