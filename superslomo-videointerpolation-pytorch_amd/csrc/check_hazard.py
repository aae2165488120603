#!/usr/bin/env python3
"""Build-time fence (called by check_isa.sh on the disassembly of every object): on gfx9 a scalar register written by a VECTOR
instruction (v_readfirstlane_b32 - what the compiler emits when a wave-uniform value lives in a vector register) needs 5 wait states
before a vector-memory instruction uses it as its scalar base.  The compiler's hazard recogniser covers its own instructions but not
the global_load_lds / global_store instructions of this library's asm statements; one such pair sent stores to a stale address
(memory access fault) when a register-allocation change moved a store base into vector registers.  The asm statements whose base may
come from a readfirstlane carry their own s_nop; this script fails the build if any vector-memory instruction still reads a scalar
register within 5 wait states of the v_readfirstlane that wrote it.
Second rule (gfx940+): a vector-memory store of more than 64 bits (dwordx3 / dwordx4) followed by a VALU write of its data registers needs
2 wait states; the epilogues' stores are opaque asm statements, so the compiler does not see the pair either.
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

def vrange(tok):
    """'v5' -> (5, 5); 'v[4:7]' -> (4, 7); anything else -> None"""
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    return (int(m.group(1)),) * 2 if m else None


WIDE_STORE = re.compile(r"\b(?:global|flat|scratch|buffer)_store_dwordx[34]\s+(.*)")
for idx, text in enumerate(ins):
    m = WIDE_STORE.search(text)
    if not m:
        continue
    ops = [o.strip() for o in m.group(1).split(",")]
    data = None
    for o in ops[1:2] if text.lstrip().startswith(("global", "flat", "scratch")) else ops[0:1]:
        data = vrange(o)
    if data is None:
        continue
    ws = 0
    for j in range(idx + 1, min(idx + 4, len(ins))):
        p = ins[j]
        if p.startswith(("s_endpgm", "s_branch", "s_setpc_b64")):          # the listing's next line is another block: nothing follows the store here
            break
        nm = re.match(r"s_nop (\d+)", p)
        if nm:
            ws += int(nm.group(1)) + 1
        else:
            if ws < 2 and p.startswith("v_"):
                dst = vrange(p.split(None, 1)[1].split(",")[0].strip()) if len(p.split(None, 1)) > 1 else None
                if dst and dst[0] <= data[1] and data[0] <= dst[1]:
                    bad += 1
                    print("check_hazard: %s: `%s` overwrites the data of `%s` %d wait state(s) after it" % (sys.argv[1], p, text, ws), file=sys.stderr)
            ws += 1
        if ws >= 2:
            break
sys.exit(1 if bad else 0)
