#!/bin/bash
# round-5 GPU call: bench lines after the U-Net kernel work (configs[3], default, fp32x3 inside default)
cd $GRAFT_REPO_ROOT
O=gpurun_out
B="python bench.py --no-cpu-baseline --no-other-configs"
$B --mlp-precision bf16 --igres 64 256 256 > $O/r5_r_c4.json 2>$O/r5_r_c4.err
$B > $O/r5_r_c2.json 2>$O/r5_r_c2.err
$B --mlp-precision bf16 > $O/r5_r_bf16c2.json 2>/dev/null
for f in r5_r_c4 r5_r_c2 r5_r_bf16c2; do python - <<PY
import json
try:
    j = json.load(open("$O/$f.json"))
    print("$f", round(j["ms_per_step"], 2), j.get("ms_per_step_fp32x3"), j["per_rank"]["unet_fwd_ms"], j["per_rank"]["unet_bwd_ms"], j["roofline"].get("frac"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
