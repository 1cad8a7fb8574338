R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/tests.log
for ov in 0 1; do
  STPDE_OVERLAP_SYNC=$ov STPDE_BENCH_ONE_DEVICE=1 STPDE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 2 --points 262144 --no-cpu-baseline > $O/bench_2rank_gloo_ov$ov.json 2> $O/bench_2rank_gloo_ov$ov.err
done
tail -3 $O/tests.log
