cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CYCLES -d /tmp/g1 -- python $R/bench.py --steps 1 --warmup 1 --points 262144 --no-cpu-baseline > /tmp/g1.log 2>&1
db=$(find /tmp/g1 -name "*.db" | head -1)
for c in GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CYCLES; do python $R/tools/rocprof_summary.py pmc $db $c | head -7; done
rocprofv3 --kernel-trace --stats -d /tmp/g2 -- python $R/bench.py --steps 1 --warmup 1 --points 262144 --no-cpu-baseline > /tmp/g2.log 2>&1
python $R/tools/rocprof_summary.py trace $(find /tmp/g2 -name "*.db" | head -1) | head -6
rocm-smi --showclocks 2>/dev/null | head -20
