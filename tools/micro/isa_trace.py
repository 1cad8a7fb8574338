"""One character per instruction of a line range of an ISA listing: M = MFMA, v = VALU, t = transcendental, d = DS read,
D = DS write, g = VMEM load, G = VMEM store, W = s_waitcnt, B = barrier, s = SALU / other, a = accvgpr move.

    python tools/micro/isa_trace.py FILE.s FIRST LAST
"""
import re
import sys


def cls(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("v_accvgpr"): return "a"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"): return "t"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "d"
    if op.startswith("ds_"): return "D"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("scratch_load"): return "g"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("scratch_store") or op.startswith("global_atomic"): return "G"
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_nop"): return "n"
    return "s"


def main():
    lines = open(sys.argv[1]).read().splitlines()
    a, b = int(sys.argv[2]), int(sys.argv[3])
    out = []
    for ln in lines[a - 1:b]:
        m = re.match(r"\s+([a-z_0-9]+)\b", ln)
        if m and not ln.strip().startswith(";"):
            out.append(cls(m.group(1)))
    s = "".join(out)
    for i in range(0, len(s), 120):
        print(s[i:i + 120])
    import collections
    print(dict(collections.Counter(s)))


if __name__ == "__main__":
    main()
