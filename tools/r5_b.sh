#!/bin/bash
# round 5, GPU call B: (3,4) stream set (configs[4]) -- parity tests, bench A/B; composite + many-rank tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py -x -q -m gpu -k "config4 or config5 or generic" > $O/c5_tests.log 2>&1; echo "c5 tests rc $?"
tail -4 $O/c5_tests.log
for v in "1 0" "1 1" "0 0"; do set -- $v
  STPDE_S34=$1 STPDE_COOP_S8_MC4=$2 timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --sub > $O/c5_s34_$1_mc4_$2.json 2> $O/c5_$1_$2.err
  python - "$O/c5_s34_$1_mc4_$2.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(j["ms_per_step"],1), j["peak_GB"], j["roofline"]["frac"], j["roofline"]["kernels"])
PY
done
timeout 1500 python -m pytest tests/test_gpu_train_loop.py -x -q -m gpu -s > $O/train_loop.log 2>&1; echo "train loop rc $?"
grep -n "composite\|passed\|failed\|Error" $O/train_loop.log | head
