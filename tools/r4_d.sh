#!/bin/bash
# round 4, call 4: k_fc1_dgrad_spec (bf16 mode) parity + timing A/B, adversarial fp32x3 product test, nccl test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4d
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_fp32x3_products.py -m gpu -q -k "nccl or fp32x3" > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log
cp gpurun_out/fp32x3_product_errors.json $O/ 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -k "bf16" > $O/pytest_bf16.log 2>&1
echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -6 $O/pytest_bf16.log
for v in 1 0; do
  STPDE_BF_SPEC_DGRAD=$v timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 > $O/bench_bf16_dspec$v.json 2> $O/bench_bf16_dspec$v.err
done
python - <<'P'
import json
for f in ("1","0"):
    try:
        d=json.load(open('gpurun_out/r4d/bench_bf16_dspec%s.json'%f)); print(f, round(d['ms_per_step'],2), d['config']['loss'], d['roofline']['kernels'])
    except Exception as e: print(f, "ERR", e)
P
