#!/bin/bash
# round-5 GPU call: U-Net weight-gradient kernels (pipelined 3x3x3 LDS kernel, 1x1x1 LDS kernel): tests + per-layer profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet3d.py tests/test_gpu_conv_fused.py tests/test_gpu_resblock_fused.py tests/test_gpu_bf16_kernel_variants.py -q -m gpu -x > gpurun_out/r5_l_tests.log 2>&1
echo "tests rc $?"; tail -3 gpurun_out/r5_l_tests.log | cut -c1-200
timeout 300 python tools/unet_profile.py 64 256 256 > gpurun_out/r5_l_unet_c4.txt 2>&1
head -3 gpurun_out/r5_l_unet_c4.txt
STPDE_CONV1_WGRAD_LDS=0 timeout 300 python tools/unet_profile.py 64 256 256 > gpurun_out/r5_l_unet_c4_no1.txt 2>&1
head -1 gpurun_out/r5_l_unet_c4_no1.txt
STPDE_CONV_WGRAD_LDS_GX=256 timeout 300 python tools/unet_profile.py 64 256 256 > gpurun_out/r5_l_unet_c4_gx256.txt 2>&1
head -1 gpurun_out/r5_l_unet_c4_gx256.txt
timeout 300 python tools/unet_profile.py 32 128 128 > gpurun_out/r5_l_unet_c2.txt 2>&1
head -1 gpurun_out/r5_l_unet_c2.txt
