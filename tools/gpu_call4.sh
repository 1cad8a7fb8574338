#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
CHECK_REPS=6 timeout 300 python tools/micro/check_fc1_fused.py 4000 > gpurun_out/c4_check.log 2>&1; echo "check rc $?"; tail -30 gpurun_out/c4_check.log | cut -c1-300
