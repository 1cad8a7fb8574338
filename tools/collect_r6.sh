# copy the outputs of tools/final_r6.sh (gpurun_out/, scratch) into profiles/ (tracked), named per round
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
for f in r6_bench r6_bench_driver_cmd r6_bench_c5 r6_bench_leakyrelu r6_bench_fp32x3 r6_bench_bf16_mode_c2grid r6_bench_config4_bf16 r6_proxy_524288 r6_proxy_262144 r6_proxy_131072 r6_bench_2rank_gloo r6_bench_train_default r6_bench_config0_gpu r6_inference r6_next_rows r6_det_cost; do
  [ -s $G/$f.json ] && cp $G/$f.json $P/$f.json
done
for f in r6_kernel_trace_stats r6_c4_kernel_trace_stats r6_c5_kernel_trace_stats r6_fp32x3_kernel_trace_stats r6_train_default_kernel_trace_stats r6_bf16_pmc_sq_counters; do
  [ -s $G/prof_r6/$f.txt ] && cp $G/prof_r6/$f.txt $P/$f.txt
done
cp $G/prof_r6/r6_pmc_FETCH_SIZE.txt $P/r6_pmc_fetch_size.txt
cp $G/prof_r6/r6_pmc_WRITE_SIZE.txt $P/r6_pmc_write_size.txt
cat $G/prof_r6/r6_pmc_SQ_BUSY_CYCLES.txt $G/prof_r6/r6_pmc_SQ_INSTS_MFMA.txt $G/prof_r6/r6_pmc_SQ_INSTS_VALU.txt $G/prof_r6/r6_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt $G/prof_r6/r6_pmc_SQ_WAVE_CYCLES.txt > $P/r6_pmc_sq_counters.txt
cp $G/prof_r6/pmc_traffic.json $P/pmc_traffic.json
python - <<PY
import json
rows = {}
for n in (524288, 262144, 131072):
    j = json.load(open("profiles/r6_proxy_%d.json" % n)); rows[str(n)] = dict(ms_per_step=j["ms_per_step"], points_per_s=j["value"])
full = json.load(open("profiles/r6_bench.json"))
json.dump(dict(note="single-GPU proxies of the per-rank work of a strong-scaling run of 2^20 points (bench.py --points N/world): what one rank of an N-GPU job computes before any exchange; NOT a measured multi-GPU curve",
               full_2p20=dict(ms_per_step=full["ms_per_step"], points_per_s=full["value"]), per_rank_points=rows,
               speedup_before_comm={k: round(full["ms_per_step"] / v["ms_per_step"], 2) for k, v in rows.items()}),
          open("profiles/r6_scaling_proxy.json", "w"), indent=1)
PY
ls $P | grep r6_ | wc -l
