#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4e
mkdir -p $O
python tools/micro/ablate_dgrad_spec.py run 2>&1 | grep -v amdgpu.ids | tee $O/ablate_dgrad_spec.txt
timeout 900 python -m pytest tests/test_gpu_lig_jet.py tests/test_gpu_reference_fixtures.py -m gpu -q -k "bf16" > $O/pytest_bf16.log 2>&1
echo "pytest rc=$?" >> $O/pytest_bf16.log
tail -4 $O/pytest_bf16.log
STPDE_BF_SPEC_DGRAD=1 timeout 600 python bench.py --no-cpu-baseline --steps 8 --warmup 3 --mlp-precision bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
python -c "
import json; d=json.load(open('$O/bench_bf16.json')); print(round(d['ms_per_step'],2), d['config']['loss'], d['roofline']['kernels'])"
