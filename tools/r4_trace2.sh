#!/bin/bash
# kernel timeline of the 2^17-point step (fused residual blocks): where the fixed per-rank cost sits
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4t
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/trp
rocprofv3 --kernel-trace --output-format csv -d /tmp/trp -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 --points 131072 > $O/trace.json 2> $O/trace.err
f=$(find /tmp/trp -name "*kernel_trace.csv" | head -1)
python - "$f" $O/timeline.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one timed step: between two consecutive k_gather_tile launches in the middle of the run
g = [i for i, r in enumerate(rows) if "k_gather_tile" in r["Kernel_Name"]]
a, b = g[3], g[4]
t0 = int(rows[a]["Start_Timestamp"])
out = open(sys.argv[2], "w")
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%9.1f %8.1f gap %7.1f q=%s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
    prev_end = max(prev_end, e)
out.write("step span %.1f us, %d kernels\n" % ((prev_end - t0) / 1e3, b - a))
P
tail -1 $O/timeline.txt
