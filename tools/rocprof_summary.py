"""Summarise rocprofv3 rocpd databases (kernel trace / PMC passes) into small text tables for profiles/.

    python tools/rocprof_summary.py trace  <results.db>            # per-kernel count / total / avg / min / max
    python tools/rocprof_summary.py pmc    <results.db> <COUNTER>  # per-kernel sum and mean of one counter
"""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"void (k_\w+)<(.*)>\(", name) or re.match(r"(k_\w+)<(.*)>", name)
    if m:
        return "%s<%s>" % (m.group(1), m.group(2).replace(" ", ""))
    return name[:90]


def trace(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e6
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("# rocprofv3 --kernel-trace summary (durations in ms); %d dispatches, %.1f ms of GPU kernel time" % (len(rows), tot))
    print("%-100s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%"))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %7d %11.3f %9.3f %9.3f %9.3f %6.2f" % (short(n), a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))


def pmc(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, counter_name, value from counters_collection where counter_name = ?" % namecol,
                     (counter,)).fetchall()
    agg = {}
    for n, _, v in rows:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += v
    print("# rocprofv3 --pmc %s summary (raw counter units: KiB for FETCH_SIZE / WRITE_SIZE)" % counter)
    print("%-100s %7s %16s %16s" % ("kernel", "calls", "sum", "mean_per_call"))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-100s %7d %16.1f %16.1f" % (short(n), a[0], a[1], a[1] / a[0]))


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
