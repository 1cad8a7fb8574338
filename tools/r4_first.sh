#!/bin/bash
# round 4, first GPU call: new tests + the U-Net-backward overlap A/B + the configs[4] workload
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_interp_generic.py tests/test_gpu_reference_fixtures.py -m gpu -x -q > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
for ov in 0 1; do
  STPDE_OVERLAP_UNET_BWD=$ov python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_ov$ov.json 2> $O/bench_ov$ov.err
  STPDE_OVERLAP_UNET_BWD=$ov python bench.py --no-cpu-baseline --steps 10 --warmup 3 --points 131072 > $O/proxy17_ov$ov.json 2> $O/proxy17_ov$ov.err
done
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4a/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), d.get('peak_GB'), d.get('recompute_steps'), d['per_rank']['compute_ms'], d['per_rank']['unet_fwd_ms'], d['per_rank']['unet_bwd_ms'])
    except Exception as e: print(f, 'ERR', e)
P
