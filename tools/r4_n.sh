#!/bin/bash
# round 4: whole GPU suite + the bench lines after the weight-gradient / epilogue work
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4n
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --points 131072 > $O/proxy17.json 2> $O/proxy17.err
python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 --igres 64 256 256 > $O/bench_c4.json 2> $O/bench_c4.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4n/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d['per_rank']['compute_ms'], d['per_rank']['unet_fwd_ms'], d['per_rank']['unet_bwd_ms']); print('   ', d['roofline']['kernels'])
    except Exception as e: print(f, 'ERR', e)
P
