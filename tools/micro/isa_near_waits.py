#!/usr/bin/env python3
"""Find vector-memory loads that were meant to be in flight during a matrix phase but are waited for right behind their issue.

Round 6 finding (k_wgrad_coop, exact fp32): the loop issued the next row tile's adjoint blocks before its MFMAs, and the listing
had `s_waitcnt vmcnt(1)` right behind them -- a register loaded in the PROLOGUE was first used at the loop head, the early loads
sat in run-time branches, so the only statically safe counter value in front of that use was "all but the last load".  Every
wave sat through a memory round trip per row tile in front of its MFMAs (14 % of the kernel).  The compiler cannot know; the
listing shows it.  This script scans every kernel of the built objects for the pattern:

    a `s_waitcnt vmcnt(N)` that has more than N vector-memory loads within the WINDOW instructions in front of it (so it waits for
    loads issued a few instructions earlier), with at least MIN_MFMA matrix instructions within the AFTER instructions behind it
    (so it sits in front of a matrix phase), inside a loop (a backward branch targets an address at or in front of it).

    python tools/micro/isa_near_waits.py [needle ...]         (kernels whose mangled name contains every needle)
"""
import glob
import os
import re
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from check_dpp_hazard import BUILD, disassemble  # noqa: E402

WINDOW, AFTER, MIN_MFMA = 48, 60, 8


def kernels(text):
    name, body = None, []
    for raw in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", raw.strip())
        if m and not (m.group(2).startswith("L") and m.group(2)[1:].isdigit()):
            if name:
                yield name, body
            name, body = m.group(2), []
            continue
        s = raw.split("//")
        ins = s[0].strip()
        if not ins or re.match(r"^[0-9a-f]+ <", ins):
            continue
        addr = int(s[1].split(":")[0].strip(), 16) if len(s) > 1 else 0
        body.append((addr, ins))
    if name:
        yield name, body


def is_load(ins):
    return ins.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")) and " lds" not in ins


def scan(body):
    # loop extents from backward branches
    loops = []
    for i, (addr, ins) in enumerate(body):
        m = re.match(r"s_cbranch_\w+ (\d+)|s_branch (\d+)", ins)
        if m:
            off = int(m.group(1) or m.group(2))
            if off >= 32768:
                tgt = addr + 4 + 4 * (off - 65536)
                loops.append((tgt, addr))
    hits = []
    for i, (addr, ins) in enumerate(body):
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
        if not m:
            continue
        n = int(m.group(1))
        if not any(lo <= addr <= hi for lo, hi in loops):
            continue
        loads = 0
        for j in range(i - 1, max(-1, i - 1 - WINDOW), -1):
            if re.match(r"s_waitcnt.*vmcnt\(0\)", body[j][1]):
                break
            loads += is_load(body[j][1])
        if loads <= n:
            continue
        mfma = sum("v_mfma" in body[j][1] for j in range(i + 1, min(len(body), i + 1 + AFTER)))
        if mfma >= MIN_MFMA:
            hits.append((addr, n, loads, mfma))
    return hits


def main():
    needles = sys.argv[1:]
    total = 0
    with tempfile.TemporaryDirectory() as scratch:
        for obj in sorted(glob.glob(os.path.join(BUILD, "*.hip.o"))):
            text = disassemble(obj, scratch)
            for name, body in kernels(text):
                if not all(n in name for n in needles):
                    continue
                for addr, n, loads, mfma in scan(body):
                    total += 1
                    print("%-24s %-90s @%x vmcnt(%d) behind %d loads, %d MFMAs follow" % (
                        os.path.basename(obj), name[:90], addr, n, loads, mfma))
    print("%d suspicious waits" % total)


if __name__ == "__main__":
    main()
