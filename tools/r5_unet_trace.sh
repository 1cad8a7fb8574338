#!/bin/bash
# kernel trace of the U-Net alone (tools/unet_profile.py), fused residual blocks on / off, both grids
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r5_unet
mkdir -p $O
export TMPDIR=/tmp
for g in "32 128 128" "64 256 256"; do
for f in ${FUSED_LIST:-1 0}; do
  tag=$(echo $g | tr ' ' 'x')_f$f
  rm -rf /tmp/tr_$tag
  STPDE_FUSED_RESBLOCK=$f rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python tools/unet_profile.py $g > $O/prof_$tag.txt 2>&1
  f2=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
  python - "$f2" > $O/kernels_$tag.txt <<'P'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    m = re.match(r"void (k_\w+)<(.*)>\(", n) or re.match(r"(k_\w+)<(.*)>", n)
    n = "%s<%s>" % (m.group(1), m.group(2).replace(" ", "")) if m else n[:80]
    n = "%s  grid=%s" % (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(a[1] for a in agg.values())
print("# %d dispatches, %.2f ms of GPU kernel time (whole profile run = 1 warm-up + 3 timed forward/backward + per-kernel section)" % (len(rows), tot))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-80s %6d %9.3f ms %8.1f us/call" % (n, a[0], a[1], 1e3 * a[1] / a[0]))
P
  head -2 $O/prof_$tag.txt | tail -1
  head -${TOPN:-16} $O/kernels_$tag.txt
done
done
