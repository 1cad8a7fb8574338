#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/c6_bench.json 2> gpurun_out/c6_bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/c6_bench.json'))
r=j['roofline']
print('ms', j['ms_per_step'], 'x3', j.get('ms_per_step_fp32x3'), 'kernel', r['kernel'], 'frac', r['frac'], 'alg', r.get('algorithmic_frac'), 'median', r.get('median_launch_ms'), 'avg', r['avg_launch_ms'])
print({k:(v.get('ms_per_step'), v.get('error')) for k,v in j['other_configs'].items()})
print('collectives', j.get('collectives_per_step'))
PY
timeout 1200 python tools/resample_stress.py --iters 1500 --out gpurun_out/c6_resample_stress.json > gpurun_out/c6_resample.log 2>&1; echo "stress rc $?"; cat gpurun_out/c6_resample.log | cut -c1-200
