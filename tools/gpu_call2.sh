#!/bin/bash
# round 6, GPU call 2: the library after the switch / dead-variant cleanup, built with compressed code objects
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c2_smoke.log 2>&1; rc=$?; echo "smoke rc $rc"; tail -3 gpurun_out/c2_smoke.log
if [ $rc -ne 0 ]; then echo "SMOKE FAILED -- stopping"; exit 1; fi
timeout 600 python bench.py --workload train_default --steps 50 > gpurun_out/c2_train_default.json 2> gpurun_out/c2_train_default.err; echo "train_default rc $?"; tail -c 1800 gpurun_out/c2_train_default.json; tail -5 gpurun_out/c2_train_default.err
timeout 600 python bench.py --workload c1 --steps 50 > gpurun_out/c2_c1.json 2> gpurun_out/c2_c1.err; echo "c1 rc $?"; tail -c 1500 gpurun_out/c2_c1.json
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/c2_tests.log 2>&1; echo "tests rc $?"
grep -n "passed\|failed\|Error" gpurun_out/c2_tests.log | tail -5; tail -30 gpurun_out/c2_tests.log | head -60
