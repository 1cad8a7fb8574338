# Round-2 profile collection (run on the GPU box through gpurun): kernel trace of the default bench command, then the
# FETCH_SIZE / WRITE_SIZE counter passes (separate runs, --pmc only) on the default 2^20-point launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $O/r2_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p_$c -name "*.db" | head -1) $c > $O/r2_pmc_$c.txt
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p_sq.log 2>&1
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p_sq -name "*.db" | head -1) $c | head -14 > $O/r2_pmc_$c.txt; done
ls -la $O
