# kernel times of the BatchNorm kernels at the two full-resolution shapes, for the row-dealing variants of the backward kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "0 1024" "1 1024" "1 2048" "0 2048" "1 4096"; do
  set -- $cfg
  for shape in "4194304 32" "4194304 16" "524288 32"; do
    rm -rf /tmp/bnp
    STPDE_BN_IL=$1 STPDE_BN_RGRID=$2 rocprofv3 --kernel-trace -d /tmp/bnp -- python $R/tools/micro/bn_time.py $shape > /dev/null 2>&1
    echo "il=$1 rgrid=$2 shape=$shape"
    python $R/tools/rocprof_summary.py trace $(find /tmp/bnp -name "*.db" | head -1) | grep k_bn | cut -c1-30,100-170
  done
done
