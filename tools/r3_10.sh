R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -x 2>&1 | tail -4 > $O/tests.log
for pe in 0 1 2; do
  STPDE_PERSIST=$pe python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_persist$pe.json 2> $O/bench_persist$pe.err
done
for pe in 0 1; do
  STPDE_PERSIST=$pe python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16_persist$pe.json 2> $O/bench_bf16_persist$pe.err
done
tail -3 $O/tests.log
