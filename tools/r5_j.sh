#!/bin/bash
# round 5, GPU call J: no XR any more (weight-gradient kernels transpose the raw-input fragments in LDS)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5j; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py tests/test_gpu_interp_generic.py tests/test_gpu_bf16_kernel_variants.py tests/test_gpu_fp32x3_products.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log | cut -c1-300
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mlp-precision fp32x3 --sub > $O/x3.json 2> $O/x3.err
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mlp-precision bf16 --sub > $O/bf16.json 2> $O/bf16.err
python - <<'PY'
import json
for f in ("bench","x3","bf16"):
    j=json.load(open("gpurun_out/r5j/%s.json"%f)); print(f, round(j["ms_per_step"],2), j["peak_GB"], j["roofline"]["kernels"], j["roofline"].get("gather_stage",{}).get("achieved"))
PY
