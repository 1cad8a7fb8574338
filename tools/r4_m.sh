#!/bin/bash
# round 4: phase swap of the bf16 weight gradient x operand-block stash: tests + 2x2 A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4m
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py -m gpu -q -k "bf16 or packed or one_call or g5b" > $O/pytest_bf16.log 2>&1
echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -4 $O/pytest_bf16.log
for sw in 1; do for f in 0; do
STPDE_WGRAD_SWAP=$sw STPDE_ACT16=$f python bench.py --no-cpu-baseline --steps 6 --warmup 2 --mlp-precision bf16 > $O/bench_bf16_s${sw}_a$f.json 2> $O/bench_bf16_s${sw}_a$f.err
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4m/*.json')):
    try:
        d=json.load(open(f)); k=d['roofline']['kernels']; print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d['config'].get('loss'), 'fc1 fwd/dgrad/wgrad', k['layer1_fwd'], k['layer1_dgrad'], k['layer1_wgrad'], 'fc2', k['layer2_fwd'], k['layer2_dgrad'], k['layer2_wgrad'])
    except Exception as e: print(f, 'ERR', e)
P
