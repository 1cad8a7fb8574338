#!/bin/bash
# round 4 (second session), baseline: whole GPU suite + the four bench lines of the round + U-Net profiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4g
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --points 131072 > $O/proxy17.json 2> $O/proxy17.err
python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 --igres 64 256 256 > $O/bench_c4.json 2> $O/bench_c4.err
python tools/unet_profile.py 32 128 128 > $O/unet_c2.txt 2>&1
python tools/unet_profile.py 64 256 256 > $O/unet_c4.txt 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4g/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d['per_rank']['compute_ms'], d['per_rank']['unet_fwd_ms'], d['per_rank']['unet_bwd_ms']); print('   ', d['roofline']['kernels'])
    except Exception as e: print(f, 'ERR', e)
P
head -3 $O/unet_c2.txt; head -3 $O/unet_c4.txt
