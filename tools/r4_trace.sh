#!/bin/bash
# kernel timeline of the 2^17-point step with / without the U-Net backward beside the IM-NET weight gradients
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4b
mkdir -p $O
python -m pytest tests/test_gpu_train_loop.py tests/test_gpu_interp_generic.py -m gpu -q > $O/pytest_new.log 2>&1
echo "pytest rc=$?" >> $O/pytest_new.log
tail -4 $O/pytest_new.log
export TMPDIR=/tmp
for ov in 1 0; do
  rm -rf /tmp/tr$ov
  STPDE_OVERLAP_UNET_BWD=$ov rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$ov -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --points 131072 > $O/trace_ov$ov.json 2> $O/trace_ov$ov.err
  f=$(find /tmp/tr$ov -name "*kernel_trace.csv" | head -1)
  python - "$f" $O/timeline_ov$ov.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cols = rows[0].keys()
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the timed steps only: the last 40 % of the trace is the profiling / side-figure part -> take a window in the middle
t0 = int(rows[0]["Start_Timestamp"])
out = open(sys.argv[2], "w")
out.write("columns: %s\n" % ",".join(cols))
n = len(rows)
for r in rows[int(n * 0.30):int(n * 0.30) + 2500]:
    out.write("%10.1f %9.1f q=%s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                         r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
P
done
python bench.py --no-cpu-baseline --steps 5 --warmup 2 --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
python -c "
import json; d=json.load(open('$O/bench_c5.json')); print('c5', d['ms_per_step'], d['peak_GB'], d['recompute_steps'], d['roofline']['kernels'])"
