#!/bin/bash
# round 4: fused residual blocks -- new tests, the U-Net tests, U-Net profiles, bench lines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4h
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resblock_fused.py -m gpu -x -q > $O/pytest_fused.log 2>&1
echo "pytest rc=$?" >> $O/pytest_fused.log
tail -30 $O/pytest_fused.log
timeout 900 python -m pytest tests/test_unet3d.py tests/test_gpu_reference_fixtures.py tests/test_gpu_train_loop.py -m gpu -q > $O/pytest_unet.log 2>&1
echo "pytest rc=$?" >> $O/pytest_unet.log
tail -15 $O/pytest_unet.log
for f in 0 1; do
STPDE_FUSED_RESBLOCK=$f python tools/unet_profile.py 32 128 128 > $O/unet_c2_f$f.txt 2>&1
STPDE_FUSED_RESBLOCK=$f python tools/unet_profile.py 64 256 256 > $O/unet_c4_f$f.txt 2>&1
head -1 $O/unet_c2_f$f.txt $O/unet_c4_f$f.txt | grep UNet
STPDE_FUSED_RESBLOCK=$f python bench.py --no-cpu-baseline --steps 10 --warmup 3 --points 131072 > $O/proxy17_f$f.json 2> $O/proxy17_f$f.err
STPDE_FUSED_RESBLOCK=$f python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 --igres 64 256 256 > $O/bench_c4_f$f.json 2> $O/bench_c4_f$f.err
done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d['per_rank']['compute_ms'], d['per_rank']['unet_fwd_ms'], d['per_rank']['unet_bwd_ms'])
    except Exception as e: print(f, 'ERR', e)
P
