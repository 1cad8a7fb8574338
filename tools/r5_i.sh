#!/bin/bash
# round 5, GPU call I: whole -m gpu suite + default bench line at the state with the fused fc1 backward / fp32x3 folds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5i; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/all_tests.log 2>&1; echo "all tests rc $?"; tail -6 $O/all_tests.log | cut -c1-300
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r5i/bench.json"))
print(j["ms_per_step"], j.get("ms_per_step_fp32x3"), j["roofline"]["frac"], j.get("roofline_fp32x3",{}).get("frac"))
for k,v in j.get("other_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("error"), v.get("dominant_kernel"), v.get("frac"), v.get("hbm_side"), v.get("mfma_side"))
PY
