R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -x -k "bf16" 2>&1 | tail -8 > $O/tests.log
for sp in 0 1; do
  STPDE_BF_SPEC=$sp python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16_spec$sp.json 2> $O/bench_bf16_spec$sp.err
done
ls $O
