R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
python tools/micro/ablate_layer.py stamp bf16 > $O/stamp_bf16.txt 2>&1
python tools/micro/ablate_layer.py stamp > $O/stamp_fp32.txt 2>&1
ls $O
