#!/bin/bash
# round-5 GPU call: SQ counters of the U-Net convolution kernels (k_conv3_lds, k_conv3d_wgrad_lds) on the configs[3] volume
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_unet
mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp -d /tmp/q$i -- python $R/tools/unet_profile.py 64 256 256 > /tmp/q$i.log 2>&1
  db=$(find /tmp/q$i -name "*.db" | head -1)
  for c in $grp; do python $R/tools/rocprof_summary.py pmc $db $c 2>&1 | grep -E "^#|k_conv3_lds|k_conv3d_wgrad_lds|k_conv1_wgrad|k_conv_fused<2,4,false" | head -16 > $O/$c.txt; done
done
timeout 600 rocprofv3 --kernel-trace -d /tmp/qt -- python $R/tools/unet_profile.py 64 256 256 > /tmp/qt.log 2>&1
python $R/tools/rocprof_summary.py trace $(find /tmp/qt -name "*.db" | head -1) | head -40 > $O/trace.txt
ls $O | wc -l
