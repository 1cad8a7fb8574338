R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_reference_fixtures.py -m gpu -q -x -k "data_parallel" 2>&1 | tail -15
