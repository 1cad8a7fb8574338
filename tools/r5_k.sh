#!/bin/bash
# round 5, GPU call K: packed-fp32 activation jets (common.h) -- fused-kernel ablations, whole test suite, bench lines of the three modes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5k; mkdir -p $O
python tools/micro/ablate_fc1_fused.py run 2>&1 | grep "variant\|stamps block 0" | tee $O/ablate_packed.txt
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mlp-precision fp32x3 --sub > $O/x3.json 2> $O/x3.err
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mlp-precision bf16 --sub > $O/bf16.json 2> $O/bf16.err
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mlp-precision bf16 --igres 64 256 256 --sub > $O/c3.json 2> $O/c3.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload c5 --sub > $O/c5.json 2> $O/c5.err
python - <<'PY'
import json
for f in ("bench","x3","bf16","c3","c5"):
    try:
        j=json.load(open("gpurun_out/r5k/%s.json"%f)); print(f, round(j["ms_per_step"],2), j["roofline"]["frac"], j["roofline"]["kernels"])
    except Exception as e: print(f, "ERR", e)
PY
timeout 2400 python -m pytest tests -q -m gpu > $O/all_tests.log 2>&1; echo "all tests rc $?"; tail -6 $O/all_tests.log | cut -c1-300
