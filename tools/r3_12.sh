R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_interp_generic.py tests/test_gpu_lig_jet.py -m gpu -q -x -k "not full_size" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gather_stage'])"
