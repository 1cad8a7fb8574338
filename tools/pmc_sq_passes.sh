cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/q$i -- python $R/bench.py --steps 1 --warmup 1 --points 262144 --no-cpu-baseline > /tmp/q$i.log 2>&1
  db=$(find /tmp/q$i -name "*.db" | head -1)
  for c in $grp; do python $R/tools/rocprof_summary.py pmc $db $c | head -12 > $R/gpurun_out/pmc_$c.txt; done
done
ls $R/gpurun_out | grep pmc_SQ | wc -l
