"""Summarise a `hipcc -Rpass-analysis=kernel-resource-usage` log: registers / scratch / occupancy per kernel.

    python tools/micro/resource_usage.py LOG [needle ...]     (only kernels whose demangled name contains every needle)
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    needles = sys.argv[2:]
    cur, res = None, {}
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(\w[\w ]*\w)(?: \[[\w/]+\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1)] = int(m.group(2))
    names = subprocess.run(["c++filt"], input="\n".join(res), capture_output=True, text=True).stdout.splitlines()
    for (k, v), name in zip(res.items(), names):
        if all(n in name for n in needles):
            print("%-110s VGPR %3d AGPR %3d scratch %4d occ %d LDS %6d" % (
                name[:110], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1),
                v.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
