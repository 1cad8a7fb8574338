#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_unet3d.py tests/test_gpu_train_loop.py -m gpu -x -q -k "deterministic or long_accumulator or config3_whole" > gpurun_out/c7_det.log 2>&1; echo "det tests rc $?"; grep -n "passed\|failed\|default mode\|configs\[3\] composite" gpurun_out/c7_det.log | tail; grep -n "Error\|assert" gpurun_out/c7_det.log | head -20
