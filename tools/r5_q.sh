#!/bin/bash
# round-5 GPU call: rewritten 3x3x3 LDS weight gradient (row staging, precomputed fragment addresses)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_unet3d.py tests/test_gpu_conv_fused.py tests/test_gpu_resblock_fused.py -q -m gpu -x > $O/r5_q_tests.log 2>&1
echo "tests rc $?"; tail -3 $O/r5_q_tests.log | cut -c1-200
timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_q_unet_c4.txt 2>&1; grep UNet3d $O/r5_q_unet_c4.txt; grep "wgrad.* 3 " $O/r5_q_unet_c4.txt | head -8
timeout 300 python tools/unet_profile.py 32 128 128 > $O/r5_q_unet_c2.txt 2>&1; grep UNet3d $O/r5_q_unet_c2.txt
