R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/tests.log
python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --igres 64 256 256 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
tail -2 $O/tests.log
