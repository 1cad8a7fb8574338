# Round-2 final measurement set (GPU box, through gpurun): profiles, then the bench lines of every mode.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
bash $R/tools/profile_r2.sh > $O/profile_r2.log 2>&1
cd $R
python tools/make_traffic_json.py $O/prof_r2/r2_pmc_FETCH_SIZE.txt $O/prof_r2/r2_pmc_WRITE_SIZE.txt 1048576 softplus $O/prof_r2/pmc_traffic.json
python bench.py --traffic-json $O/prof_r2/pmc_traffic.json > $O/r2_bench.json 2> $O/r2_bench.err
python bench.py --act leakyrelu --no-cpu-baseline > $O/r2_bench_leakyrelu.json 2> $O/r2_bench_leakyrelu.err
python bench.py --mlp-precision fp32x3 --no-cpu-baseline > $O/r2_bench_fp32x3.json 2> $O/r2_bench_fp32x3.err
python bench.py --mlp-precision bf16 --no-cpu-baseline > $O/r2_bench_bf16_mode_c2grid.json 2> /dev/null
python bench.py --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > $O/r2_bench_config4_bf16.json 2> /dev/null
for p in 524288 262144 131072; do python bench.py --points $p --steps 8 --warmup 2 --no-cpu-baseline > $O/r2_proxy_$p.json 2> /dev/null; done
python tools/bench_inference.py > $O/r2_inference.json 2> /dev/null
python tools/bench_inference.py --mlp-precision fp32x3 > $O/r2_inference_fp32x3.json 2> /dev/null
for f in r2_bench r2_bench_leakyrelu r2_bench_fp32x3 r2_bench_bf16_mode_c2grid r2_bench_config4_bf16 r2_proxy_524288 r2_proxy_262144 r2_proxy_131072; do python - <<PY
import json
j = json.load(open("$O/$f.json"))
print("$f", round(j["value"]), round(j["ms_per_step"], 2), j["roofline"].get("frac"), j["roofline"].get("step_frac_per_gpu"))
PY
done
