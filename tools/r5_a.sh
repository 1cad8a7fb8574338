#!/bin/bash
# round 5, GPU call A: new tests first, then the whole -m gpu suite, then the default bench line (with other_configs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5a; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_resblock_fused.py -x -q -m gpu > $O/new_tests.log 2>&1; echo "new tests rc $?" 
tail -5 $O/new_tests.log
timeout 1200 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r5a/bench.json"))
print(j["ms_per_step"], j.get("ms_per_step_fp32x3"), j.get("roofline_fp32x3",{}).get("kernels"))
for k,v in j.get("other_configs",{}).items(): print(k, v.get("ms_per_step"), v.get("error"), v.get("dominant_kernel"), v.get("frac"))
PY
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_train_loop.py --deselect tests/test_gpu_resblock_fused.py > $O/all_tests.log 2>&1; echo "all tests rc $?"
tail -5 $O/all_tests.log
python tools/unet_profile.py 64 256 256 > $O/r5_unet_profile_c4.txt 2>/dev/null
python tools/unet_profile.py 32 128 128 > $O/r5_unet_profile_c2.txt 2>/dev/null
head -12 $O/r5_unet_profile_c4.txt
