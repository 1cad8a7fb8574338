# Round-5 closing measurement set (GPU box, through gpurun) after the last kernel change of the round (k_layer_coop: first-pass
# barrier, epilogue operand prefetch): the lines and profiles that change with it.  The U-Net / bf16-counter / inference files of
# tools/final_r5.sh are not re-taken (their kernels did not change).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
P=$O/prof_r5
mkdir -p $P
NB="--no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 $NB > $P/bench_under_rocprof.json 2> /tmp/kt.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt -name "*.db" | head -1) > $P/r5_kernel_trace_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/p_$c -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_$c.log 2>&1
  python $R/tools/rocprof_summary.py pmc $(find /tmp/p_$c -name "*.db" | head -1) $c > $P/r5_pmc_$c.txt
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/p_sq -- python $R/bench.py --steps 1 --warmup 1 $NB > /tmp/p_sq.log 2>&1
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES; do python $R/tools/rocprof_summary.py pmc $(find /tmp/p_sq -name "*.db" | head -1) $c | head -16 > $P/r5_pmc_$c.txt; done
rocprofv3 --kernel-trace --stats -d /tmp/ktx -- python $R/bench.py --steps 3 --warmup 1 --mlp-precision fp32x3 $NB > $P/bench_x3_under_rocprof.json 2> /tmp/ktx.log
python $R/tools/rocprof_summary.py trace $(find /tmp/ktx -name "*.db" | head -1) > $P/r5_fp32x3_kernel_trace_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/kt5 -- python $R/bench.py --steps 2 --warmup 1 --workload c5 $NB > $P/bench_c5_under_rocprof.json 2> /tmp/kt5.log
python $R/tools/rocprof_summary.py trace $(find /tmp/kt5 -name "*.db" | head -1) > $P/r5_c5_kernel_trace_stats.txt
cd $R
python tools/make_traffic_json.py $P/r5_pmc_FETCH_SIZE.txt $P/r5_pmc_WRITE_SIZE.txt 1048576 softplus $P/pmc_traffic.json
python bench.py --traffic-json $P/pmc_traffic.json > $O/r5_bench.json 2> $O/r5_bench.err
python bench.py --act leakyrelu --no-cpu-baseline --no-other-configs > $O/r5_bench_leakyrelu.json 2> /dev/null
python bench.py --mlp-precision fp32x3 --no-cpu-baseline > $O/r5_bench_fp32x3.json 2> /dev/null
python bench.py --mlp-precision bf16 --no-cpu-baseline > $O/r5_bench_bf16_mode_c2grid.json 2> /dev/null
python bench.py --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > $O/r5_bench_config4_bf16.json 2> /dev/null
for p in 524288 262144 131072; do python bench.py --points $p --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r5_proxy_$p.json 2> /dev/null; done
STPDE_BENCH_ONE_DEVICE=1 STPDE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/r5_bench_2rank_gloo.json 2> /dev/null
python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/r5_bench_c5.json 2> /dev/null
python tools/bench_inference.py > $O/r5_inference.json 2> /dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r5_bench_driver_cmd.json 2> /dev/null
for f in r5_bench r5_bench_driver_cmd r5_bench_c5 r5_bench_leakyrelu r5_bench_fp32x3 r5_bench_bf16_mode_c2grid r5_bench_config4_bf16 r5_proxy_524288 r5_proxy_262144 r5_proxy_131072 r5_bench_2rank_gloo; do python - <<PY
import json
try:
    j = json.load(open("$O/$f.json"))
    print("$f", round(j["value"]), round(j["ms_per_step"], 2), j["roofline"].get("frac"), j["roofline"].get("step_frac_per_gpu"), j.get("ms_per_step_fp32x3"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
