#!/bin/bash
# round-5 GPU call: LDS-tiled 3x3x3 convolution (k_conv3_lds): tests + per-layer profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_fused.py tests/test_unet3d.py tests/test_gpu_resblock_fused.py -q -m gpu -x > $O/r5_n_tests.log 2>&1
echo "tests rc $?"; tail -3 $O/r5_n_tests.log | cut -c1-200
timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_n_unet_c4.txt 2>&1; grep UNet3d $O/r5_n_unet_c4.txt
STPDE_CONV3_LDS=0 timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_n_unet_c4_off.txt 2>&1; grep UNet3d $O/r5_n_unet_c4_off.txt
timeout 300 python tools/unet_profile.py 32 128 128 > $O/r5_n_unet_c2.txt 2>&1; grep UNet3d $O/r5_n_unet_c2.txt
STPDE_CONV3_LDS=0 timeout 300 python tools/unet_profile.py 32 128 128 > $O/r5_n_unet_c2_off.txt 2>&1; grep UNet3d $O/r5_n_unet_c2_off.txt
