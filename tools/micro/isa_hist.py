"""Instruction-class histogram per basic block of one kernel in a hipcc -S listing (gfx950).

usage: isa_hist.py file.s <mangled-name-substring> [--top N] [--dump BLOCK]
Classes: mfma, valu (v_* except mfma), trans (v_exp/log/rcp/rsq/sqrt), accmov (v_accvgpr_*), lds (ds_*), vmem
(global_/buffer_/scratch_), salu (s_* except waitcnt/barrier/nop), wait (s_waitcnt, s_nop, s_barrier).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accmov"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", op):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, needle = sys.argv[1], sys.argv[2]
    top = 12
    dump = None
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
    if "--dump" in sys.argv:
        dump = sys.argv[sys.argv.index("--dump") + 1]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and needle in l and l.split(":")[0].endswith("LayerArgs") or (
            l.startswith("_Z") and needle in l and ":" in l
        ):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1 :]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s)
    print(lines[start].split(":")[0])
    tot = collections.Counter()
    rows = []
    for name, ins in blocks.items():
        c = collections.Counter(classify(x.split()[0]) for x in ins)
        tot.update(c)
        back = [x for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        rows.append((name, len(ins), c, back))
    rows_sorted = sorted(rows, key=lambda r: -r[1])[:top]
    keys = ["mfma", "valu", "trans", "accmov", "lds", "vmem", "salu", "wait"]
    print("%-12s %6s " % ("block", "n") + " ".join("%6s" % k for k in keys) + "  branches")
    for name, n, c, back in rows_sorted:
        print("%-12s %6d " % (name, n) + " ".join("%6d" % c[k] for k in keys) + "  " + ",".join(b.split()[-1] for b in back))
    print("%-12s %6d " % ("TOTAL", sum(tot.values())) + " ".join("%6d" % tot[k] for k in keys))
    if dump:
        ops = collections.Counter(x.split()[0] for x in blocks[dump])
        for k, v in ops.most_common(60):
            print("   %-40s %d" % (k, v))


if __name__ == "__main__":
    main()
