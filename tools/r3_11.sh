R=$GRAFT_REPO_ROOT
cd $R
for t in stamp prio1 prio2; do python tools/micro/ablate_layer.py stamp_spec $t 2>&1 | grep -v amdgpu | head -3; done
