#!/usr/bin/env python3
"""Scan the gfx950 code objects of the built library for data hazards the compiler does not guard: DPP reads of freshly
written registers inside asm statements (rule 1 below), and 16-byte buffer stores whose data registers are overwritten by the
very next instruction (rule 2, scan_store_data).

Rule (gfx9 / CDNA data-hazard table): a DPP instruction must not read a VGPR that a VALU instruction wrote fewer than TWO wait
states earlier; the hardware does not interlock.  The compiler's hazard recognizer inserts the s_nop itself for DPP instructions
it emits, but it does not parse inline assembly, so hand-written `v_add_f32_dpp` sequences (csrc/common.h: row_sum16 /
row_sum16x4) need their own wait states.  ADVICE round 5 found 4 + sites where a DPP add read a register written one
instruction earlier; this scan is the build check that keeps them from coming back (run by __graft_entry__.build() and by
tests/test_host_logic.py).

Method: the fat binary of every object file under csrc/build is un-bundled (and inflated) in a scratch directory, its gfx950
code object disassembled, and every `*_dpp` instruction checked against the straight-line instructions in front of it: each
instruction between the writer and the DPP reader counts one wait state, `s_nop N` counts N + 1.  Labels (branch targets) end the
backward scan -- a hazard across a taken branch is not seen (none of the hand-written DPP code sits at a block entry: each
asm block opens with its own s_nop).

Exit status 0 = clean; prints the offending sites otherwise.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "..", "space_time_pde_amd", "csrc", "build")

_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _split(line):
    """mnemonic, [operands] of one disassembly line ('\\tv_add_f32_dpp v1, v1, v1 row_shr:1 ... // 0000...')"""
    line = line.split("//")[0].strip()
    if not line:
        return None, []
    parts = line.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return mn, ops


def _valu_writes(mn, ops):
    """VGPRs a VALU instruction writes (first operand; compares / readlane write scalar registers)."""
    if not mn.startswith("v_") or not ops:
        return set()
    if mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_mfma", "v_smfmac")):
        return set()      # (MFMA results have their own, compiler-visible, much longer dependency rules)
    return _regs(ops[0])


def scan_disassembly(text, need=2):
    """-> list of (kernel, dpp line, writer line, wait states seen)"""
    bad = []
    kernel = "?"
    window = []            # (mnemonic, ops, raw) of the straight-line code in front
    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw.strip())
        if m:
            name = m.group(1)
            if not name.startswith("L") or not name[1:].isdigit():
                kernel = name
            window = []
            continue
        mn, ops = _split(raw)
        if mn is None or not raw.startswith(("\t", " ")):
            continue
        if mn.endswith("_dpp") or " dpp8:" in raw or "row_shr:" in raw or "row_shl:" in raw or "quad_perm:" in raw \
                or "row_bcast:" in raw or "wave_shr:" in raw or "row_ror:" in raw or "row_mirror" in raw \
                or "row_half_mirror" in raw or "row_newbcast:" in raw:
            reads = set()
            for o in ops[1:]:
                reads |= _regs(o.split()[0] if o.split() else o)
            reads |= _regs(ops[0]) if ops else set()       # the old destination is read under row / bank masks
            ws = 0
            for pmn, pops, praw in reversed(window):
                if ws >= need:
                    break
                if pmn == "s_nop":
                    ws += int(pops[0], 0) + 1 if pops else 1
                    continue
                hit = _valu_writes(pmn, pops) & reads
                if hit:
                    bad.append((kernel, raw.strip().split("//")[0].strip(), praw.strip().split("//")[0].strip(), ws))
                    break
                ws += 1
        window.append((mn, ops, raw))
        if len(window) > 16:
            window.pop(0)
    return bad


def scan_store_data(text, need=2):
    """Second rule (round 6, found on hardware with k_fc1_bwd_fused): `buffer_store_dwordx3/x4 ... sN offen` followed by a VALU
    write of one of its DATA registers with no wait state in between stored the NEW value of the first data dword now and then
    (gfx950, two waves per SIMD; the compiler's hazard model exempts buffer stores whose soffset is a scalar register, so it puts
    nothing in between -- tools/micro/check_fc1_fused.py: 2,300 of 6.1 M tangent row sums wrong and different from run to run, 0
    with `s_nop 2` behind the store).  Flag every such store whose data is overwritten fewer than `need` wait states later.
    -> list of (kernel, store line, writer line, wait states)"""
    bad, kernel = [], "?"
    lines = text.splitlines()
    for i, raw in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw.strip())
        if m:
            if not (m.group(1).startswith("L") and m.group(1)[1:].isdigit()):
                kernel = m.group(1)
            continue
        s = raw.split("//")[0].strip()
        if not (s.startswith("buffer_store_dwordx4") or s.startswith("buffer_store_dwordx3")):
            continue
        data = _regs(s.split(",")[0])
        ws = 0
        for k in range(1, need + 2):
            if i + k >= len(lines) or ws >= need:
                break
            n = lines[i + k].split("//")[0].strip()
            mn, ops = _split(lines[i + k])
            if mn is None or re.match(r"^[0-9a-f]+ <", n):
                break
            if mn == "s_nop":
                ws += int(ops[0], 0) + 1 if ops else 1
                continue
            if _valu_writes(mn, ops) & data:
                bad.append((kernel, s, n, ws))
                break
            ws += 1
    return bad


def disassemble(obj, scratch):
    """gfx950 disassembly text of one host object with an embedded offload bundle ('' if it has none).  The fat binary section
    is dumped with llvm-objcopy and un-bundled with clang-offload-bundler, which also inflates the compressed bundles of
    --offload-compress builds (llvm-objdump --offloading hands those out still compressed)."""
    base = os.path.join(scratch, os.path.basename(obj))
    fat, co = base + ".fat", base + ".co"
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, base + ".tmp"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    text = ""
    if r.returncode == 0 and os.path.exists(fat):
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--input=" + fat, "--unbundle",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, check=False)
        if r.returncode == 0 and os.path.exists(co):
            r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, check=False)
            text = r.stdout.decode(errors="replace")
    for f in glob.glob(base + "*"):
        os.remove(f)
    return text


def scan_build(build_dir=BUILD, verbose=False):
    objs = sorted(glob.glob(os.path.join(build_dir, "*.hip.o")))
    bad, ndpp = [], 0
    with tempfile.TemporaryDirectory() as scratch:
        for o in objs:
            text = disassemble(o, scratch)
            n = len(re.findall(r"_dpp\b", text))
            ndpp += n
            b = scan_disassembly(text) + scan_store_data(text)
            if verbose:
                print("%-28s %6d dpp instructions, %d hazards" % (os.path.basename(o), n, len(b)))
            bad += [(os.path.basename(o),) + x for x in b]
    return bad, ndpp, len(objs)


if __name__ == "__main__":
    bad, ndpp, nobj = scan_build(verbose=True)
    for b in bad[:40]:
        print("HAZARD %s  %s\n    reader: %s\n    writer: %s   (%d wait states)" % b)
    print("%d objects, %d DPP instructions, %d hazards" % (nobj, ndpp, len(bad)))
    sys.exit(1 if bad else 0)
