# Round-5 final measurement set (GPU box, through gpurun): profiles, then the bench lines of every mode.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/profile_r5.sh > $O/profile_r5.log 2>&1
bash $R/tools/micro/pmc_bf16.sh > $O/pmc_bf16_r5.log 2>&1
cd $R
python tools/make_traffic_json.py $O/prof_r5/r5_pmc_FETCH_SIZE.txt $O/prof_r5/r5_pmc_WRITE_SIZE.txt 1048576 softplus $O/prof_r5/pmc_traffic.json
python tools/make_traffic_json.py $O/prof_r5/r5_c4_pmc_FETCH_SIZE.txt $O/prof_r5/r5_c4_pmc_WRITE_SIZE.txt 1048576 softplus $O/prof_r5/pmc_traffic_c4_bf16.json bf16
python bench.py --traffic-json $O/prof_r5/pmc_traffic.json > $O/r5_bench.json 2> $O/r5_bench.err
python bench.py --act leakyrelu --no-cpu-baseline --no-other-configs > $O/r5_bench_leakyrelu.json 2> $O/r5_bench_leakyrelu.err
python bench.py --mlp-precision fp32x3 --no-cpu-baseline > $O/r5_bench_fp32x3.json 2> $O/r5_bench_fp32x3.err
python bench.py --mlp-precision bf16 --no-cpu-baseline > $O/r5_bench_bf16_mode_c2grid.json 2> /dev/null
python bench.py --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline --traffic-json $O/prof_r5/pmc_traffic_c4_bf16.json > $O/r5_bench_config4_bf16.json 2> $O/r5_bench_config4_bf16.err
for p in 524288 262144 131072; do python bench.py --points $p --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r5_proxy_$p.json 2> /dev/null; done
STPDE_BENCH_ONE_DEVICE=1 STPDE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > $O/r5_bench_2rank_gloo.json 2> /dev/null
python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/r5_bench_c5.json 2> /dev/null
python tools/bench_inference.py > $O/r5_inference.json 2> /dev/null
python tools/bench_next_rows.py > $O/r5_next_rows.json 2> /dev/null
python tools/unet_profile.py 64 256 256 > $O/r5_unet_profile_c4.txt 2> /dev/null
python tools/unet_profile.py 32 128 128 > $O/r5_unet_profile_c2.txt 2> /dev/null
FUSED_LIST=1 bash tools/r5_unet_trace.sh > $O/r5_unet_trace.log 2>&1
bash tools/r5_o.sh > $O/r5_unet_pmc.log 2>&1
cd $R
for f in r5_bench r5_bench_c5 r5_bench_leakyrelu r5_bench_fp32x3 r5_bench_bf16_mode_c2grid r5_bench_config4_bf16 r5_proxy_524288 r5_proxy_262144 r5_proxy_131072 r5_bench_2rank_gloo; do python - <<PY
import json
try:
    j = json.load(open("$O/$f.json"))
    print("$f", round(j["value"]), round(j["ms_per_step"], 2), j["roofline"].get("frac"), j["roofline"].get("step_frac_per_gpu"), j.get("ms_per_step_fp32x3"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
