#!/bin/bash
# round 4: operand-block stash (act16) of the first hidden layer -- bf16 tests + A/B bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4k
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py -m gpu -q -k "bf16 or packed or one_call or g5b" > $O/pytest_bf16.log 2>&1
echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -6 $O/pytest_bf16.log
for f in 0 1; do
STPDE_ACT16=$f python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 > $O/bench_bf16_a$f.json 2> $O/bench_bf16_a$f.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4k/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d['config'].get('loss')); print('   ', d['roofline']['kernels'])
    except Exception as e: print(f, 'ERR', e)
P
