#!/bin/bash
# round 5, GPU call 5: two-rank bench, parity suites, division self-test, fast-division A/B, replay profile, bench lines
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( INFX_DIST_BACKEND=gloo MASTER_PORT=29641 INFX_THREADS=4 timeout 600 python bench.py --gpus 2 --docs 140000 --steps 2 --warmup 1 --batch 200 --no-cpu-baseline > $O/bench_w2.json 2> $O/bench_w2.err; echo "rc=$?" >> $O/bench_w2.err )
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_lookups.py -m gpu -q > $O/gputest.log 2>&1; echo "gputest rc=$?" >> $O/gputest.log )
for v in main fastdiv; do
  L=$GRAFT_REPO_ROOT/infidex_amd/libinfidex_hip_$v.so; [ $v = main ] && L=$GRAFT_REPO_ROOT/infidex_amd/libinfidex_hip.so
  INFX_LIB=$L timeout 300 python bench.py --steps 8 --warmup 2 --sessions 1 --no-cpu-baseline > $O/acc_$v.json 2> $O/acc_$v.err
done
INFX_EXACT_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --sessions 1 --no-cpu-baseline > $O/prof_reg.json 2> $O/prof_reg.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
grep -v "^\s*$" $O/bench_w2.err | grep -v "Warning\|amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*\|Gloo" | tail -8; head -c 400 $O/bench_w2.json; echo
tail -12 $O/gputest.log
grep -h "replayed queries of 1000" -A 22 $O/prof_reg.err | head -24
for f in acc_main acc_fastdiv prof_reg bench_20; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f p95 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms'], d['p95_batch_latency_ms']), 'acc %.3f' % d['roofline']['avg_launch_ms'], [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d.get('roofline_by_kernel',[])])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
