# Round-4 profile collection (GPU box, through gpurun): kernel trace of the default bench command, FETCH_SIZE / WRITE_SIZE and
# SQ counter passes (separate runs, --pmc only) on the default 2^20-point launch; the same for BASELINE configs[3] (bf16).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r4
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $O/r4_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p_$c -name "*.db" | head -1) $c > $O/r4_pmc_$c.txt
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p_sq.log 2>&1
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p_sq -name "*.db" | head -1) $c | head -16 > $O/r4_pmc_$c.txt; done
# configs[3]: latent [1,64,256,256,32], bf16 MFMA operands
C4="--mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python $R/bench.py --steps 2 --warmup 1 $C4 > $O/bench_c4_under_rocprof.json 2> /tmp/kt4.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt4 -name "*.db" | head -1) > $O/r4_c4_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p4_$c -- python $R/bench.py --steps 1 --warmup 1 $C4 > /tmp/p4_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p4_$c -name "*.db" | head -1) $c > $O/r4_c4_pmc_$c.txt
done
ls -la $O
