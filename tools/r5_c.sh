#!/bin/bash
# round 5, GPU call C: fused fc1 backward (bf16 mode) -- parity A/B, bf16 tests, bench A/B on both grids
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16_kernel_variants.py -x -q -m gpu -k fused_fc1 > $O/ab_test.log 2>&1; echo "A/B test rc $?"; tail -15 $O/ab_test.log
timeout 1500 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py -x -q -m gpu -k "bf16 or full_size or g5b" > $O/bf16_tests.log 2>&1; echo "bf16 tests rc $?"; tail -4 $O/bf16_tests.log
for f in 1 0; do
  STPDE_FC1_FUSED=$f timeout 600 python bench.py --mlp-precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/bf16_c2grid_f$f.json 2> $O/bf16_f$f.err
  python - "$O/bf16_c2grid_f$f.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j["ms_per_step"],2), j["roofline"]["kernels"])
PY
done
STPDE_FC1_FUSED=1 timeout 600 python bench.py --mlp-precision bf16 --igres 64 256 256 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/c3_f1.json 2> $O/c3_f1.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r5c/c3_f1.json")); print("configs[3]", round(j["ms_per_step"],2), j["roofline"]["kernels"])
PY
