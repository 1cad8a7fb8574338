# round 3, first GPU call: full -m gpu tests, default bench line, configs[3] (bf16) bench + kernel trace + SQ counters
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/tests.log
python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
python bench.py --steps 5 --warmup 2 --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 2 --warmup 1 --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > /tmp/kt.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $O/c4_kernel_trace.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d /tmp/q$i -- python $R/bench.py --steps 1 --warmup 1 --points 262144 --mlp-precision bf16 --no-cpu-baseline > /tmp/q$i.log 2>&1
  db=$(find /tmp/q$i -name "*.db" | head -1)
  for c in $grp; do python $R/tools/rocprof_summary.py pmc $db $c | head -16 > $O/c4pmc_$c.txt; done
done
ls $O
