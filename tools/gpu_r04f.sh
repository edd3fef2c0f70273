#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fullsize.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/summary.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" >> $O/summary.txt; }
run base A=1
run base2 A=1
cat $O/summary.txt; tail -5 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04f/*.json')):
    d=json.load(open(f)); r=d['roofline']
    print(f.split('/')[-1], round(d['value']), 'ms/step %.2f p50 %.1f' % (d['ms_per_step'], d['p50_batch_latency_ms']), 'acc %.3f' % r['avg_launch_ms'], {k: round(v,2) for k,v in r['other_kernels_ms'].items()},
          [(x['kernel'], round(x['avg_launch_ms'],2)) for x in d.get('roofline_by_kernel',[])], r['replay_flag_reasons_per_launch'], r['exact_replays_per_launch'])
PY
