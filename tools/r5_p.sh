#!/bin/bash
# round-5 GPU call: k_conv3_lds epilogue / phase-offset variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_fused.py tests/test_unet3d.py tests/test_gpu_resblock_fused.py -q -m gpu -x > $O/r5_p_tests.log 2>&1
echo "tests rc $?"; tail -3 $O/r5_p_tests.log | cut -c1-200
for ph in default 0 2 8; do
  if [ $ph = default ]; then unset STPDE_CONV3_LDS_PHASE; else export STPDE_CONV3_LDS_PHASE=$ph; fi
  timeout 300 python tools/unet_profile.py 64 256 256 > $O/r5_p_unet_c4_ph$ph.txt 2>&1; echo "phase $ph"; grep UNet3d $O/r5_p_unet_c4_ph$ph.txt; grep "conv3d_fused.* 3 " $O/r5_p_unet_c4_ph$ph.txt | head -4
done
unset STPDE_CONV3_LDS_PHASE
timeout 300 python tools/unet_profile.py 32 128 128 > $O/r5_p_unet_c2.txt 2>&1; grep UNet3d $O/r5_p_unet_c2.txt
