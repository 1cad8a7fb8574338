# round 3, second GPU call: new tests, N1 timing, bf16 8-wave shape A/B, single-pass fc3 weight gradient A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
python -m pytest tests/test_next_rows.py -m gpu -q -x 2>&1 | tail -5 > $O/tests.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err
for sh in 0 28; do
  STPDE_BF_SHAPE=$sh python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16_shape$sh.json 2> $O/bench_bf16_shape$sh.err
done
STPDE_BF_SHAPE=28 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -k "bf16" 2>&1 | tail -5 > $O/tests_bf28.log
for m in 0 1; do
  STPDE_WGRAD_MCW4=$m python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_mcw4_$m.json 2> $O/bench_mcw4_$m.err
done
STPDE_WGRAD_MCW4=1 python -m pytest tests/test_gpu_reference_fixtures.py -m gpu -q -k "g5b or benchmarked" 2>&1 | tail -5 > $O/tests_mcw4.log
ls $O
