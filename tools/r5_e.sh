#!/bin/bash
# round 5, GPU call E: fused fc1 backward (no-spill version): direct comparison, ablations, parity tests, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5f; mkdir -p $O
python tools/micro/check_fc1_fused.py 1000 2>&1 | grep -v amdgpu.ids | tee $O/check.txt
python tools/micro/check_fc1_fused.py 77 2>&1 | grep -v amdgpu.ids | tee -a $O/check.txt
echo "--- old = cooperative kernel (STPDE_BF_SPEC_DGRAD=0)" | tee -a $O/check.txt
STPDE_BF_SPEC_DGRAD=0 python tools/micro/check_fc1_fused.py 1000 2>&1 | grep -v amdgpu.ids | tee -a $O/check.txt
python tools/micro/ablate_fc1_fused.py run 2>&1 | grep variant | tee $O/ablate_fc1_fused.txt
timeout 900 python -m pytest tests/test_gpu_bf16_kernel_variants.py tests/test_gpu_reference_fixtures.py tests/test_gpu_lig_jet.py -x -q -m gpu -k "fused_fc1 or g5b or fp32x3 or bf16" > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
STPDE_FC1_FUSED=1 timeout 600 python bench.py --mlp-precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --sub > $O/bf16_c2grid_f1.json 2> $O/bf16_f1.err
python - <<'PY'
import json
for f in ("bf16_c2grid_f1",):
    j=json.load(open("gpurun_out/r5f/%s.json"%f)); print(f, round(j["ms_per_step"],2), j["roofline"]["kernels"])
PY
