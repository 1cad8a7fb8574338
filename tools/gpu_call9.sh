#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_unet3d.py tests/test_gpu_train_loop.py -m gpu -x -q -k "deterministic or long_accumulator or config3_whole" > gpurun_out/c9_det.log 2>&1; echo "det tests rc $?"; grep -n "passed\|failed" gpurun_out/c9_det.log | tail -3; grep -n "^E " gpurun_out/c9_det.log | head -20
timeout 600 python tools/det_cost.py > gpurun_out/c9_det_cost.json 2> gpurun_out/c9_det_cost.err; echo "det cost rc $?"; cat gpurun_out/c9_det_cost.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c9_tests.log 2>&1; echo "tests rc $?"; grep -n "passed\|failed" gpurun_out/c9_tests.log | tail -3; grep -n "^E " gpurun_out/c9_tests.log | head -20
