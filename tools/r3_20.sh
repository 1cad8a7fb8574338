R=$GRAFT_REPO_ROOT
cd $R
for L in A B A B; do
  STPDE_LIB=$R/tools/micro/_abl/lib$L.so python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']; print('$L', round(d['ms_per_step'],2), k['layer1_wgrad'], k['layer2_wgrad'], k['layer1_dgrad'], k['layer1_fwd'])"
done
