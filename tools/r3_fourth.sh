# round 3, fourth GPU call: full tests with the one-call-per-direction path, step time with / without it at 2^17 and 2^20 points
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/tests.log
for pipe in 0 1; do
  STPDE_PIPELINE=$pipe python bench.py --steps 10 --warmup 3 --points 131072 --no-cpu-baseline > $O/bench_p17_pipe$pipe.json 2> $O/bench_p17_pipe$pipe.err
done
STPDE_PIPELINE=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_pipe1.json 2> $O/bench_pipe1.err
ls $O
