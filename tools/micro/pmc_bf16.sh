cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bf16_r5
mkdir -p $O
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --mlp-precision bf16"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pb_$i -- $B > /tmp/pb_$i.log 2>&1
  for c in $set; do python $R/tools/rocprof_summary.py pmc $(find /tmp/pb_$i -name "*.db" | head -1) $c | head -14 > $O/$c.txt; done
done
ls $O
