cd $GRAFT_REPO_ROOT
O=gpurun_out
( time python -m pytest tests -m gpu -x -q ) > $O/v2_tests.log 2>&1
python bench.py > $O/v2_bench.json 2> $O/v2_bench.err
grep -E "passed|failed" $O/v2_tests.log | tail -2; cut -c1-400 $O/v2_bench.json
