#!/bin/bash
# round-5 GPU call: 64-channel LDS weight gradient, grid policy of the LDS weight gradients below 2 M voxels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_unet3d.py tests/test_gpu_conv_fused.py tests/test_gpu_resblock_fused.py -q -m gpu -x > $O/r5_m_tests.log 2>&1
echo "tests rc $?"; tail -3 $O/r5_m_tests.log | cut -c1-200
timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_m_unet_c4.txt 2>&1; grep UNet3d $O/r5_m_unet_c4.txt
STPDE_CONV_WGRAD_LDS_HALF_BELOW=0 timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_m_unet_c4_full.txt 2>&1; grep UNet3d $O/r5_m_unet_c4_full.txt
STPDE_CONV_WGRAD_LDS_HALF_BELOW=0 timeout 300 python tools/unet_profile.py 32 128 128 > $O/r5_m_unet_c2_full.txt 2>&1; grep UNet3d $O/r5_m_unet_c2_full.txt
B="python bench.py --no-cpu-baseline --no-other-configs --steps 6 --warmup 2"
$B --mlp-precision bf16 --igres 64 256 256 > $O/r5_m_c4.json 2>/dev/null
STPDE_CONV_WGRAD_LDS_HALF_BELOW=0 $B --mlp-precision bf16 --igres 64 256 256 > $O/r5_m_c4_full.json 2>/dev/null
$B --points 131072 > $O/r5_m_p17.json 2>/dev/null
STPDE_CONV_WGRAD_LDS_HALF_BELOW=0 $B --points 131072 > $O/r5_m_p17_full.json 2>/dev/null
$B --mlp-precision bf16 > $O/r5_m_bf16c2.json 2>/dev/null
STPDE_CONV_WGRAD_LDS_HALF_BELOW=0 $B --mlp-precision bf16 > $O/r5_m_bf16c2_full.json 2>/dev/null
for f in r5_m_c4 r5_m_c4_full r5_m_p17 r5_m_p17_full r5_m_bf16c2 r5_m_bf16c2_full; do python - <<PY
import json
try:
    j = json.load(open("$O/$f.json"))
    print("$f", round(j["ms_per_step"], 2), j["per_rank"]["unet_fwd_ms"], j["per_rank"]["unet_bwd_ms"])
except Exception as e:
    print("$f", "ERR", e)
PY
done
