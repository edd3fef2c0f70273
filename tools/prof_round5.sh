#!/bin/bash
# Round-5 profile on the GPU box: full GPU suite + smoke, bench lines (driver's 20-step form with the CPU baseline, 256 steps, sessions 3 / 5, configs 2 / 3 / 5, the sharded
# path on one rank), kernel trace, PMC passes (each its own run; --pmc never together with trace flags) for k_accumulate AND the replay kernels.  Output: gpurun_out/prof_r05/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05; rm -rf $O; mkdir -p $O
( cd $R && timeout 900 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log )
python $R/bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
python $R/bench.py --no-cpu-baseline > $O/bench_256steps.json 2> $O/bench_256steps.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sessions 3 > $O/bench_20steps_sessions3.json 2> $O/bench_20steps_sessions3.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sessions 5 > $O/bench_20steps_sessions5.json 2> $O/bench_20steps_sessions5.err
python $R/bench.py --config 2 --steps 60 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python $R/bench.py --config 3 --steps 60 --warmup 5 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python $R/bench.py --config 5 --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
INFX_FORCE_SHARDED=1 python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
B="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --sessions 1"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- $B > $O/kt.json 2> $O/kt.err
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/kt
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt3 -o kt --output-format csv -- python $R/bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --sessions 1 > $O/kt_cfg3.json 2> $O/kt_cfg3.err
find $O/kt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_cfg3.csv; rm -rf $O/kt3
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p --output-format csv -- $B > $O/pmc_sq.json 2> $O/pmc_sq.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o p --output-format csv -- $B > $O/pmc_sq2.json 2> $O/pmc_sq2.err
for d in pmc_fetch pmc_sq pmc_sq2; do
  python $R/tools/pmc_summary.py $O/$d "k_accumulate<8192, 2>" > $O/$d.txt 2>&1
  python $R/tools/pmc_summary.py $O/$d "k_ex_" > $O/${d}_replay.txt 2>&1
  rm -rf $O/$d
done
python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_sha16())" > $O/kernel_sha16.txt
tail -3 $O/gputest.log; tail -2 $O/smoke.log
cat $O/pmc_fetch.txt $O/pmc_sq.txt | head -30; cat $O/pmc_sq_replay.txt $O/pmc_sq2_replay.txt | head -90; head -24 $O/kernel_stats.csv | cut -c1-50,150-
for f in bench_20steps bench_256steps bench_20steps_sessions3 bench_20steps_sessions5 bench_cfg2 bench_cfg3 bench_cfg5 bench_sharded_w1; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), (d.get('cpu_baseline') or {}).get('identical_topk_sets'), (d.get('cpu_baseline') or {}).get('value'), {k: round(v,2) for k,v in d['stage_ms_per_step'].items() if k.startswith('plan')})
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
