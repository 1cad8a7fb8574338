#!/bin/bash
# round 5, GPU call D: fused fc1 backward v3 timing + ablations; fp32x3 with the folded raw-input tiles
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5d; mkdir -p $O
python tools/micro/ablate_fc1_fused.py run 2>&1 | grep variant | tee $O/ablate_fc1_fused.txt
timeout 900 python -m pytest tests/test_gpu_bf16_kernel_variants.py tests/test_gpu_reference_fixtures.py -x -q -m gpu -k "fused_fc1 or g5b or fp32x3" > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
STPDE_FC1_FUSED=1 timeout 600 python bench.py --mlp-precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/bf16_c2grid_f1.json 2> $O/bf16_f1.err
timeout 600 python bench.py --mlp-precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/x3.json 2> $O/x3.err
STPDE_X3_XFOLD=0 STPDE_X3_FC2_QUAD=0 timeout 600 python bench.py --mlp-precision fp32x3 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/x3_old.json 2> $O/x3_old.err
python - <<'PY'
import json
for f in ("bf16_c2grid_f1","x3","x3_old"):
    j=json.load(open("gpurun_out/r5d/%s.json"%f)); print(f, round(j["ms_per_step"],2), j["roofline"]["kernels"])
PY
