#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/det_cost.py > gpurun_out/c8_det_cost.json 2> gpurun_out/c8_det_cost.err; echo "det cost rc $?"; cat gpurun_out/c8_det_cost.json; tail -3 gpurun_out/c8_det_cost.err
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c8_tests.log 2>&1; echo "tests rc $?"; grep -n "passed\|failed" gpurun_out/c8_tests.log | tail -3; grep -n "^E " gpurun_out/c8_tests.log | head -20
