R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q 2>&1 | grep -E "passed|failed" > $O/tests.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
cat $O/tests.log
