# copy the outputs of tools/final_r5.sh (gpurun_out/, scratch) into profiles/ (tracked), named per round
set -e
cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
for f in r5_bench r5_bench_c5 r5_bench_leakyrelu r5_bench_fp32x3 r5_bench_bf16_mode_c2grid r5_bench_config4_bf16 r5_proxy_524288 r5_proxy_262144 r5_proxy_131072 r5_bench_2rank_gloo r5_inference r5_next_rows; do
  [ -s $G/$f.json ] && cp $G/$f.json $P/$f.json
done
for f in r5_unet_profile_c4 r5_unet_profile_c2; do [ -s $G/$f.txt ] && grep -v amdgpu.ids $G/$f.txt > $P/$f.txt; done
cp $G/prof_r5/r5_kernel_trace_stats.txt $P/r5_kernel_trace_stats.txt
cp $G/prof_r5/r5_c4_kernel_trace_stats.txt $P/r5_c4_kernel_trace_stats.txt
cp $G/prof_r5/r5_c5_kernel_trace_stats.txt $P/r5_c5_kernel_trace_stats.txt
cp $G/prof_r5/r5_fp32x3_kernel_trace_stats.txt $P/r5_fp32x3_kernel_trace_stats.txt
for f in $G/r5_unet/kernels_*.txt; do [ -s $f ] && cp $f $P/r5_unet_$(basename $f); done
cp $G/prof_r5/r5_pmc_FETCH_SIZE.txt $P/r5_pmc_fetch_size.txt
cp $G/prof_r5/r5_pmc_WRITE_SIZE.txt $P/r5_pmc_write_size.txt
cp $G/prof_r5/r5_c4_pmc_FETCH_SIZE.txt $P/r5_c4_pmc_fetch_size.txt
cp $G/prof_r5/r5_c4_pmc_WRITE_SIZE.txt $P/r5_c4_pmc_write_size.txt
cat $G/prof_r5/r5_pmc_SQ_BUSY_CYCLES.txt $G/prof_r5/r5_pmc_SQ_INSTS_MFMA.txt $G/prof_r5/r5_pmc_SQ_INSTS_VALU.txt $G/prof_r5/r5_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt $G/prof_r5/r5_pmc_SQ_WAVE_CYCLES.txt > $P/r5_pmc_sq_counters.txt
cp $G/prof_r5/pmc_traffic.json $P/pmc_traffic.json
cp $G/prof_r5/pmc_traffic_c4_bf16.json $P/pmc_traffic_c4_bf16.json
# SQ counters of the bf16 mode (tools/micro/pmc_bf16.sh)
: > $P/r5_bf16_pmc_sq_counters.txt
for f in $G/prof_bf16_r5/*.txt; do cat $f >> $P/r5_bf16_pmc_sq_counters.txt; done
# SQ counters of the U-Net convolution kernels on the configs[3] volume (tools/r5_o.sh)
: > $P/r5_unet_pmc_sq_counters.txt
for c in GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM; do [ -s $G/pmc_unet/$c.txt ] && cat $G/pmc_unet/$c.txt >> $P/r5_unet_pmc_sq_counters.txt; done
[ -s $G/pmc_unet/trace.txt ] && cp $G/pmc_unet/trace.txt $P/r5_unet_pmc_kernel_trace.txt
python - <<PY
import json
rows = {}
for n in (524288, 262144, 131072):
    j = json.load(open("profiles/r5_proxy_%d.json" % n)); rows[str(n)] = dict(ms_per_step=j["ms_per_step"], points_per_s=j["value"])
full = json.load(open("profiles/r5_bench.json"))
json.dump(dict(note="single-GPU proxies of the per-rank work of a strong-scaling run of 2^20 points (bench.py --points N/world): what one rank of an N-GPU job computes before any exchange; NOT a measured multi-GPU curve",
               full_2p20=dict(ms_per_step=full["ms_per_step"], points_per_s=full["value"]), per_rank_points=rows,
               speedup_before_comm={k: round(full["ms_per_step"] / v["ms_per_step"], 2) for k, v in rows.items()}),
          open("profiles/r5_scaling_proxy.json", "w"), indent=1)
PY
ls -la $P | wc -l
