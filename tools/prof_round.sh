#!/bin/bash
# Round profile on the GPU box: default bench line (with the CPU baseline), kernel trace, PMC passes for k_accumulate (each its own run; --pmc never
# together with trace flags).  Output: gpurun_out/prof_r03/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r03; rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $O/pmc_tcc -o p --output-format csv -- $B > $O/pmc_tcc.json 2> $O/pmc_tcc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p --output-format csv -- $B > $O/pmc_sq.json 2> $O/pmc_sq.err
for d in pmc_fetch pmc_tcc pmc_sq; do python $R/tools/pmc_summary.py $O/$d "k_accumulate<8192, 2>" > $O/$d.txt 2>&1; rm -rf $O/$d; done
sha256sum $R/infidex_amd/csrc/stage1.hip.inc | cut -c1-16 > $O/kernel_sha16.txt
for st in 2 8; do INFX_ACC_STRIPE=$st timeout 300 python $R/bench.py --steps 30 --warmup 4 --no-cpu-baseline > $O/bench_stripe$st.json 2> $O/bench_stripe$st.err; done
cat $O/pmc_fetch.txt $O/pmc_tcc.txt $O/pmc_sq.txt; head -14 $O/kernel_stats.csv | cut -c1-160; tail -c 1500 $O/bench_default.json
