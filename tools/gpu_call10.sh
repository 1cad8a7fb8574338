#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/c10_bench.json 2> gpurun_out/c10_bench.err; echo "bench rc $?"
timeout 900 python bench.py --igres 64 256 256 --mlp-precision bf16 --no-cpu-baseline --no-other-configs --sub --steps 8 --warmup 2 > gpurun_out/c10_bench_c4.json 2> gpurun_out/c10_bench_c4.err; echo "c4 rc $?"
python - <<'PY'
import json
for f in ('c10_bench','c10_bench_c4'):
    j=json.load(open('gpurun_out/%s.json'%f))
    print(f, j['ms_per_step'], j.get('ms_per_step_fp32x3'), j['roofline']['kernels'])
PY
timeout 600 python tools/det_cost.py > gpurun_out/c10_det_cost.json 2> gpurun_out/c10_det_cost.err; echo "det cost rc $?"; cat gpurun_out/c10_det_cost.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c10_tests.log 2>&1; echo "tests rc $?"; grep -n "passed\|failed" gpurun_out/c10_tests.log | tail -3; grep -n "^E " gpurun_out/c10_tests.log | head -20
