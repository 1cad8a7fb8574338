# Round-5 profile collection (GPU box, through gpurun): kernel trace of the default bench command, FETCH_SIZE / WRITE_SIZE and
# SQ counter passes (separate runs, --pmc only) on the default 2^20-point launch; the same for BASELINE configs[3] (bf16) and a
# kernel trace of configs[4] and of the fp32x3 line.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r5
mkdir -p $O
NB="--no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 $NB > $O/bench_under_rocprof.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $O/r5_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p_$c -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p_$c -name "*.db" | head -1) $c > $O/r5_pmc_$c.txt
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_sq.log 2>&1
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p_sq -name "*.db" | head -1) $c | head -16 > $O/r5_pmc_$c.txt; done
# fp32x3 (second headline-grade line): kernel trace
rocprofv3 --kernel-trace --stats -d /tmp/ktx -- python $R/bench.py --steps 3 --warmup 1 --mlp-precision fp32x3 $NB > $O/bench_x3_under_rocprof.json 2> /tmp/ktx.log
python $R/tools/rocprof_summary.py trace $(find /tmp/ktx -name "*.db" | head -1) > $O/r5_fp32x3_kernel_trace_stats.txt
# configs[3]: latent [1,64,256,256,32], bf16 MFMA operands
C4="--mlp-precision bf16 --igres 64 256 256 $NB"
rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python $R/bench.py --steps 2 --warmup 1 $C4 > $O/bench_c4_under_rocprof.json 2> /tmp/kt4.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt4 -name "*.db" | head -1) > $O/r5_c4_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p4_$c -- python $R/bench.py --steps 1 --warmup 1 $C4 > /tmp/p4_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p4_$c -name "*.db" | head -1) $c > $O/r5_c4_pmc_$c.txt
done
# configs[4]: user-string equations, (3,4) stream set
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -- python $R/bench.py --steps 2 --warmup 1 --workload c5 $NB > $O/bench_c5_under_rocprof.json 2> /tmp/kt5.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt5 -name "*.db" | head -1) > $O/r5_c5_kernel_trace_stats.txt
ls -la $O
