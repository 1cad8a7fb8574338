# copy the outputs of tools/final_r3.sh (gpurun_out/, scratch) into profiles/ (tracked), named per round
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
for f in r3_bench r3_bench_leakyrelu r3_bench_fp32x3 r3_bench_bf16_mode_c2grid r3_bench_config4_bf16 r3_proxy_524288 r3_proxy_262144 r3_proxy_131072 r3_bench_2rank_gloo r3_inference r3_next_rows; do
  [ -s $G/$f.json ] && cp $G/$f.json $P/$f.json
done
for f in r3_unet_profile_c4 r3_unet_profile_c2; do [ -s $G/$f.txt ] && grep -v amdgpu.ids $G/$f.txt > $P/$f.txt; done
cp $G/prof_r3/r3_kernel_trace_stats.txt $P/r3_kernel_trace_stats.txt
cp $G/prof_r3/r3_c4_kernel_trace_stats.txt $P/r3_c4_kernel_trace_stats.txt
cp $G/prof_r3/r3_pmc_FETCH_SIZE.txt $P/r3_pmc_fetch_size.txt
cp $G/prof_r3/r3_pmc_WRITE_SIZE.txt $P/r3_pmc_write_size.txt
cp $G/prof_r3/r3_c4_pmc_FETCH_SIZE.txt $P/r3_c4_pmc_fetch_size.txt
cp $G/prof_r3/r3_c4_pmc_WRITE_SIZE.txt $P/r3_c4_pmc_write_size.txt
cat $G/prof_r3/r3_pmc_SQ_BUSY_CYCLES.txt $G/prof_r3/r3_pmc_SQ_INSTS_MFMA.txt $G/prof_r3/r3_pmc_SQ_INSTS_VALU.txt $G/prof_r3/r3_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt $G/prof_r3/r3_pmc_SQ_WAVE_CYCLES.txt > $P/r3_pmc_sq_counters.txt
cp $G/prof_r3/pmc_traffic.json $P/pmc_traffic.json
cp $G/prof_r3/pmc_traffic_c4_bf16.json $P/pmc_traffic_c4_bf16.json
# SQ counters of the bf16 mode (tools/micro/pmc_bf16.sh)
: > $P/r3_bf16_pmc_sq_counters.txt
for f in $G/prof_bf16/*.txt; do cat $f >> $P/r3_bf16_pmc_sq_counters.txt; done
python - <<PY
import json
rows = {}
for n in (524288, 262144, 131072):
    j = json.load(open("profiles/r3_proxy_%d.json" % n)); rows[str(n)] = dict(ms_per_step=j["ms_per_step"], points_per_s=j["value"])
full = json.load(open("profiles/r3_bench.json"))
json.dump(dict(note="single-GPU proxies of the per-rank work of a strong-scaling run of 2^20 points (bench.py --points N/world): what one rank of an N-GPU job computes before any exchange; NOT a measured multi-GPU curve",
               full_2p20=dict(ms_per_step=full["ms_per_step"], points_per_s=full["value"]), per_rank_points=rows,
               speedup_before_comm={k: round(full["ms_per_step"] / v["ms_per_step"], 2) for k, v in rows.items()}),
          open("profiles/r3_scaling_proxy.json", "w"), indent=1)
PY
ls -la $P | wc -l
