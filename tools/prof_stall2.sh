#!/bin/bash
# instruction mix / stall counters of k_accumulate and k_accumulate3 with and without their list loops.  Output: gpurun_out/stall/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stall; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --sessions 1"
run() { tag=$1; kn=$2; shift; shift; env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/$tag -o p --output-format csv -- $B > $O/$tag.json 2> $O/$tag.err
  echo "== $tag"; python $R/tools/pmc_summary.py $O/$tag "$kn" | tee $O/$tag.txt; rm -rf $O/$tag; }
run v1_full "k_accumulate<8192, 2>" INFX_ACC_V3=0
run v1_nolist "k_accumulate<8192, 2>" INFX_ACC_V3=0 INFX_ACC_SKIP=1
run v3_full "k_accumulate3<8192, 2>" INFX_ACC_V3=1
run v3_noterm "k_accumulate3<8192, 2>" INFX_ACC_V3=1 INFX_ACC_DBG=32
run v3_nolist "k_accumulate3<8192, 2>" INFX_ACC_V3=1 INFX_ACC_DBG=16
