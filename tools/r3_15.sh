R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -x 2>&1 | tail -6 > $O/tests.log
STPDE_PIPELINE=0 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -x -k "bf16" 2>&1 | tail -4 > $O/tests_py.log
for pk in 0 1; do
  STPDE_PACKED_STASH=$pk python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16_pk$pk.json 2> $O/bench_bf16_pk$pk.err
done
tail -3 $O/tests.log; tail -2 $O/tests_py.log
