#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4l
mkdir -p $O
export TMPDIR=/tmp
rm -rf /tmp/trl
rocprofv3 --kernel-trace --output-format csv -d /tmp/trl -- python bench.py --no-cpu-baseline --steps 3 --warmup 2 --mlp-precision bf16 > $O/bench.json 2> $O/bench.err
f2=$(find /tmp/trl -name "*kernel_trace.csv" | head -1)
python - "$f2" > $O/kernels.txt <<'P'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    m = re.match(r"void (k_\w+)<(.*)>\(", n) or re.match(r"(k_\w+)<(.*)>", n)
    n = "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else n[:80]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-80s %6d %9.3f ms %8.3f ms/call" % (n, a[0], a[1], a[1] / a[0]))
P
cat $O/kernels.txt
