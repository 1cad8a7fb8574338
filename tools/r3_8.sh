R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
python tools/micro/ablate_layer.py stamp_spec > $O/stamp_spec.txt 2>&1
python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -x -k "bf16" 2>&1 | tail -4 > $O/tests.log
python bench.py --steps 4 --warmup 2 --mlp-precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
cat $O/stamp_spec.txt; cat $O/tests.log
