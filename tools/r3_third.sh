# round 3, third GPU call: bf16 ablations of the first-layer kernels, U-Net convolution counters at configs[3] size
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
python -m pytest tests/test_next_rows.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.log
python tools/micro/ablate_layer.py run bf16 > $O/ablate_bf16.txt 2>&1
python tools/micro/ablate_layer.py run > $O/ablate_fp32.txt 2>&1
python tools/unet_profile.py 64 256 256 > $O/unet_c4.txt 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/u$i -- python $R/tools/unet_profile.py 64 256 256 > /tmp/u$i.log 2>&1
  db=$(find /tmp/u$i -name "*.db" | head -1)
  for c in $grp; do python $R/tools/rocprof_summary.py pmc $db $c | head -14 > $O/unetpmc_$c.txt; done
done
ls $O
