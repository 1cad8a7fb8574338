# Round-6 measurement set (GPU box, through gpurun): kernel traces, FETCH_SIZE / WRITE_SIZE and SQ counter passes (separate runs,
# --pmc only) of the default bench command; kernel traces of fp32x3, configs[3], configs[4] and of the reference's own training
# regime (eager); the bench lines of every mode; the launch-bound workloads; deterministic-mode cost; lattice inference.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
P=$O/prof_r6
mkdir -p $P
NB="--no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 $NB > $P/bench_under_rocprof.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $P/r6_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p_$c -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p_$c -name "*.db" | head -1) $c > $P/r6_pmc_$c.txt
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_sq.log 2>&1
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p_sq -name "*.db" | head -1) $c | head -16 > $P/r6_pmc_$c.txt; done
rocprofv3 --kernel-trace --stats -d /tmp/ktx -- python $R/bench.py --steps 3 --warmup 1 --mlp-precision fp32x3 $NB > $P/bench_x3_under_rocprof.json 2> /tmp/ktx.log
python $R/tools/rocprof_summary.py trace $(find /tmp/ktx -name "*.db" | head -1) > $P/r6_fp32x3_kernel_trace_stats.txt
C4="--mlp-precision bf16 --igres 64 256 256 $NB"
rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python $R/bench.py --steps 2 --warmup 1 $C4 > $P/bench_c4_under_rocprof.json 2> /tmp/kt4.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt4 -name "*.db" | head -1) > $P/r6_c4_kernel_trace_stats.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p4_sq -- python $R/bench.py --steps 1 --warmup 1 $C4 > /tmp/p4_sq.log 2>&1
: > $P/r6_bf16_pmc_sq_counters.txt
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p4_sq -name "*.db" | head -1) $c | head -16 >> $P/r6_bf16_pmc_sq_counters.txt; done
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -- python $R/bench.py --steps 2 --warmup 1 --workload c5 $NB > $P/bench_c5_under_rocprof.json 2> /tmp/kt5.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt5 -name "*.db" | head -1) > $P/r6_c5_kernel_trace_stats.txt
# the reference's own training regime, eager launches (--profile-only: three eager iterations behind the warm-up)
rocprofv3 --kernel-trace --stats -d /tmp/ktd -- python $R/bench.py --workload train_default --profile-only > /tmp/ktd.json 2> /tmp/ktd.log
python $R/tools/rocprof_summary.py trace $(find /tmp/ktd -name "*.db" | head -1) > $P/r6_train_default_kernel_trace_stats.txt
cd $R
python tools/make_traffic_json.py $P/r6_pmc_FETCH_SIZE.txt $P/r6_pmc_WRITE_SIZE.txt 1048576 softplus $P/pmc_traffic.json
python bench.py --traffic-json $P/pmc_traffic.json > $O/r6_bench.json 2> $O/r6_bench.err
python bench.py --act leakyrelu --no-cpu-baseline --no-other-configs > $O/r6_bench_leakyrelu.json 2> /dev/null
python bench.py --mlp-precision fp32x3 --no-cpu-baseline > $O/r6_bench_fp32x3.json 2> /dev/null
python bench.py --mlp-precision bf16 --no-cpu-baseline > $O/r6_bench_bf16_mode_c2grid.json 2> /dev/null
python bench.py --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > $O/r6_bench_config4_bf16.json 2> /dev/null
for p in 524288 262144 131072; do python bench.py --points $p --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r6_proxy_$p.json 2> /dev/null; done
STPDE_BENCH_ONE_DEVICE=1 STPDE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/r6_bench_2rank_gloo.json 2> /dev/null
python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/r6_bench_c5.json 2> /dev/null
python bench.py --workload train_default --steps 100 > $O/r6_bench_train_default.json 2> /dev/null
python bench.py --workload c1 --steps 100 > $O/r6_bench_config0_gpu.json 2> /dev/null
python tools/bench_inference.py > $O/r6_inference.json 2> /dev/null
python tools/bench_next_rows.py > $O/r6_next_rows.json 2> /dev/null
python tools/det_cost.py > $O/r6_det_cost.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_driver_cmd.json 2> /dev/null
for f in r6_bench r6_bench_driver_cmd r6_bench_c5 r6_bench_leakyrelu r6_bench_fp32x3 r6_bench_bf16_mode_c2grid r6_bench_config4_bf16 r6_proxy_524288 r6_proxy_262144 r6_proxy_131072 r6_bench_2rank_gloo r6_bench_train_default r6_bench_config0_gpu; do python - <<PY
import json
try:
    j = json.load(open("$O/$f.json"))
    print("$f", round(j["value"]), round(j["ms_per_step"], 2), j["roofline"].get("kernel"), j["roofline"].get("frac"), j["roofline"].get("algorithmic_frac"), j["roofline"].get("step_frac_per_gpu"), j.get("ms_per_step_fp32x3"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
head -12 $P/r6_kernel_trace_stats.txt
