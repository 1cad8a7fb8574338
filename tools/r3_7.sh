R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
mkdir -p $O
cd $R
python tools/micro/ablate_layer.py stamp_spec > $O/stamp_spec.txt 2>&1
cat $O/stamp_spec.txt
