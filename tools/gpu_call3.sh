#!/bin/bash
# round 6, GPU call 3: k_fc1_bwd_fused v2 (four workgroups per row tile, two per CU) + launch-bound workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bf16_kernel_variants.py tests/test_gpu_reference_fixtures.py tests/test_gpu_lig_jet.py -m gpu -x -q -k "bf16 or fused or fc1" > gpurun_out/c3_tests_bf16.log 2>&1; echo "bf16 tests rc $?"; grep -n "passed\|failed" gpurun_out/c3_tests_bf16.log | tail -3
timeout 900 python bench.py --igres 64 256 256 --mlp-precision bf16 --no-cpu-baseline --no-other-configs --sub --steps 8 --warmup 2 > gpurun_out/c3_bench_c4.json 2> gpurun_out/c3_bench_c4.err; echo "c4 rc $?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/c3_bench_c4.json'))
print('configs[3] ms', j['ms_per_step'], j['roofline']['kernel'], j['roofline']['kernels'])
PY
timeout 600 python bench.py --workload train_default --steps 50 > gpurun_out/c3_train_default.json 2> gpurun_out/c3_train_default.err; echo "train_default rc $?"; tail -c 1500 gpurun_out/c3_train_default.json; tail -3 gpurun_out/c3_train_default.err | cut -c1-300
timeout 600 python bench.py --workload c1 --steps 50 > gpurun_out/c3_c1.json 2> gpurun_out/c3_c1.err; echo "c1 rc $?"; tail -c 1200 gpurun_out/c3_c1.json
