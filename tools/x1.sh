cd $GRAFT_REPO_ROOT
O=gpurun_out
T="base sb nw8o2 nw8o3 nw8o4"
python tools/micro/ablate_layer.py run2 $T > $O/x2_fc2_fp32.txt 2>&1
tail -n 6 $O/x2_*.txt
