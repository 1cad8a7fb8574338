cd $GRAFT_REPO_ROOT
echo "bf16 (row 'baseline' = committed kernels, row 'no activation jet' = early-issue dgrad loads + z0 prefetch)"
python tools/micro/ablate_layer.py run bf16 2>&1 | grep -v amdgpu
echo fp32
python tools/micro/ablate_layer.py run 2>&1 | grep -v amdgpu
