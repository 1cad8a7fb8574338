#!/bin/bash
# round 4, call 3: latency-hiding variants of the exact-fp32 tail / fc2 input-gradient kernels (A/B against the round-3 library),
# the fp32x3-parametrised parity tests, the nccl world-1 test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4c
mkdir -p $O
python -m pytest tests/test_gpu_train_loop.py -m gpu -q -k "nccl" > $O/pytest_nccl.log 2>&1; tail -3 $O/pytest_nccl.log
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py tests/test_next_rows.py -m gpu -q > $O/pytest_parity.log 2>&1
echo "pytest rc=$?" >> $O/pytest_parity.log
tail -15 $O/pytest_parity.log
for lib in new r3; do
  L=""
  [ $lib = r3 ] && L=$PWD/tools/micro/_abl/libstpde_r3.so
  STPDE_LIB=$L python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_$lib.json 2> $O/bench_$lib.err
done
python - <<'P'
import json
for f in ("new","r3"):
    d=json.load(open('gpurun_out/r4c/bench_%s.json'%f)); print(f, round(d['ms_per_step'],2), d['roofline']['kernels'])
P
