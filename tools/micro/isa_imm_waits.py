#!/usr/bin/env python3
"""List the counter waits of a kernel's loops that cover a vector-memory instruction issued a few instructions earlier.

Companion of isa_near_waits.py (round 6).  For every `s_waitcnt vmcnt(N)` inside a loop it counts the vector-memory loads / stores /
atomics among the 14 instructions in front of it; if there are more than N, the wait covers one that was only just issued -- a
load consumed at once (an exposed L2 / HBM round trip unless another wave covers it), a store whose data registers are
overwritten (the wait is for its acknowledgement), or a result register re-used by the allocator.  Prints the instruction
waited for, how far back it was issued and what follows the wait; reading the hits is manual (DESIGN 8.0 lists what they were
in the hot kernels: dead result registers of 16-byte loads, loop-carried scalars copied behind a prefetch, the last k-tile of the
weight ring, store data read late).

    python tools/micro/isa_imm_waits.py OBJECT.hip.o MANGLED_PREFIX [MANGLED_PREFIX ...]      (objects under csrc/build)
"""
import re,sys,os,tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, HERE)
from check_dpp_hazard import disassemble, BUILD
import isa_near_waits as nw
obj,needles=sys.argv[1],sys.argv[2:]
with tempfile.TemporaryDirectory() as t:
    text=disassemble(os.path.join(BUILD,obj),t)
for name,body in nw.kernels(text):
    if not any(name.startswith(n) for n in needles): continue
    loops=[]
    for i,(addr,ins) in enumerate(body):
        m=re.match(r"s_cbranch_\w+ (\d+)|s_branch (\d+)",ins)
        if m:
            off=int(m.group(1) or m.group(2))
            if off>=32768: loops.append((addr+4+4*(off-65536),addr))
    print("==",name[:80],len(body),"instr")
    for i,(addr,ins) in enumerate(body):
        m=re.match(r"s_waitcnt.*vmcnt\((\d+)\)",ins)
        if not m: continue
        if not any(lo<=addr<=hi for lo,hi in loops): continue
        n=int(m.group(1))
        vm=[j for j in range(max(0,i-14),i) if re.match(r"(global|buffer)_(load|store|atomic)",body[j][1])]
        if len(vm)>n:
            first=vm[len(vm)-n-1] if n<len(vm) else vm[0]
            nxt=[body[j][1].split()[0] for j in range(i+1,min(len(body),i+4))]
            print("  @%x vmcnt(%d): %d vmem ops in the 14 instr before; waits for '%s' issued %d instr earlier; next: %s"%(addr,n,len(vm),body[first][1][:38],i-first,' '.join(nxt)))
