#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/_build/bench_mfma_slice
$B > $O/mfma_slice.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $O/pmc -o p --output-format csv -- $B > $O/pmc.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/pmc "k_mfma_slice" > $O/pmc.txt 2>&1
rm -rf $O/pmc
cat $O/mfma_slice.txt $O/pmc.txt
