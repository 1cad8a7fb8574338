cd $GRAFT_REPO_ROOT
O=gpurun_out
for i in 1 2 3 4; do
python -m pytest tests -m gpu -x -q > $O/flake_$i.log 2>&1
grep -E "passed|failed" $O/flake_$i.log | tail -1
grep -E "AssertionError|pool " $O/flake_$i.log | head -5
done
