#!/bin/bash
# round 6, GPU call 1: baseline after the DPP fix (tests, bench, resample stress)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/c1_tests.log
tail -5 gpurun_out/c1_tests.log
timeout 600 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/c1_bench.json
timeout 900 python tools/resample_stress.py --iters 250 --out gpurun_out/c1_resample_stress.json > gpurun_out/c1_resample.log 2>&1; echo "stress rc $?"
tail -3 gpurun_out/c1_resample.log
