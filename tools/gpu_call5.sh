#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
CHECK_REPS=4 timeout 300 python tools/micro/check_fc1_fused.py 4000 > gpurun_out/c5_check.log 2>&1; echo "check rc $?"; tail -6 gpurun_out/c5_check.log | cut -c1-200
timeout 600 python -X faulthandler bench.py --workload train_default --steps 50 > gpurun_out/c5_train_default.json 2> gpurun_out/c5_train_default.err; echo "train_default rc $?"; tail -c 1600 gpurun_out/c5_train_default.json; tail -12 gpurun_out/c5_train_default.err | cut -c1-200
timeout 600 python bench.py --workload c1 --steps 50 > gpurun_out/c5_c1.json 2> gpurun_out/c5_c1.err; echo "c1 rc $?"; tail -c 1200 gpurun_out/c5_c1.json
timeout 900 python -m pytest tests/test_gpu_bf16_kernel_variants.py tests/test_gpu_reference_fixtures.py -m gpu -x -q -k "bf16 or fused or fc1" > gpurun_out/c5_tests_bf16.log 2>&1; echo "bf16 tests rc $?"; grep -n "passed\|failed" gpurun_out/c5_tests_bf16.log | tail -3
