#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5g; mkdir -p $O
python tools/micro/ablate_fc1_fused.py run 2>&1 | grep "variant\|stamps" | tee $O/ablate3.txt
FC1F_TAG=noslp python tools/micro/ablate_fc1_fused.py run 2>&1 | grep "variant\|stamps" | tee -a $O/ablate3.txt
