#!/usr/bin/env python3
"""Build-time fence (called by check_isa.sh on the disassembly of every object): on gfx9 a scalar register written by a VECTOR
instruction (v_readfirstlane_b32 - what the compiler emits when a wave-uniform value lives in a vector register) needs 5 wait states
before a vector-memory instruction uses it as its scalar base.  The compiler's hazard recogniser covers its own instructions but not
the global_load_lds / global_store instructions of this library's asm statements; one such pair sent stores to a stale address
(memory access fault) when a register-allocation change moved a store base into vector registers.  The asm statements whose base may
come from a readfirstlane carry their own s_nop; this script fails the build if any vector-memory instruction still reads a scalar
register within 5 wait states of the v_readfirstlane that wrote it.
usage: llvm-objdump -d dev.co | check_hazard.py <name>"""
import re
import sys

VMEM = re.compile(r"\b(global_load_lds_dword\w*|global_store_dword\w*|global_load_dword\w*|global_atomic_\w+|buffer_\w+)\s+(.*)")
ins = []
for line in sys.stdin:
    m = re.match(r"\s+([a-z_0-9]+\b.*?)\s*//", line)
    if m:
        ins.append(m.group(1).strip())
bad = 0
for idx, text in enumerate(ins):
    m = VMEM.search(text)
    if not m:
        continue
    sm = re.search(r"s\[(\d+):(\d+)\]", m.group(2))
    if not sm:
        continue
    regs = (int(sm.group(1)), int(sm.group(2)))
    ws = 0
    for j in range(idx - 1, max(idx - 12, -1), -1):
        p = ins[j]
        if re.match(r"(s_branch|s_setpc|s_swappc|s_endpgm)\b", p):
            break          # unconditional jump: the code behind it is reached through a label only - its predecessors are not in this listing order
                           # (a conditional branch is scanned across: the fall-through path is the straight line)
        rf = re.match(r"v_readfirstlane_b32 s(\d+),", p)
        if rf and regs[0] <= int(rf.group(1)) <= regs[1]:          # any register of the s[lo:hi] operand (a 4-SGPR buffer descriptor too)
            if ws < 5:
                bad += 1
                print("check_hazard: %s: `%s` reads s%s %d wait state(s) after `%s`" % (sys.argv[1], text, rf.group(1), ws, p), file=sys.stderr)
            break
        nm = re.match(r"s_nop (\d+)", p)
        ws += int(nm.group(1)) + 1 if nm else 1
sys.exit(1 if bad else 0)
