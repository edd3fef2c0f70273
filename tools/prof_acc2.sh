#!/bin/bash
# k_accumulate2 vs k_accumulate on the GPU box: PMC passes (each its own run; --pmc never together with trace flags).  Output under gpurun_out/prof_acc2/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_acc2; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sessions 1"
export INFX_EXACT=${INFX_EXACT:-0}
for v in 0 1; do
  export INFX_ACC_V1=$v
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $O/pmc1_v$v -o p --output-format csv -- $B > $O/pmc1_v$v.json 2> $O/pmc1_v$v.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/pmc2_v$v -o p --output-format csv -- $B > $O/pmc2_v$v.json 2> $O/pmc2_v$v.err
  rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU -d $O/pmc3_v$v -o p --output-format csv -- $B > $O/pmc3_v$v.json 2> $O/pmc3_v$v.err
  for p in 1 2 3; do python $R/tools/pmc_summary.py $O/pmc${p}_v$v k_accumulate > $O/pmc${p}_v$v.txt 2>&1; rm -rf $O/pmc${p}_v$v; done
done
cat $O/*.txt
