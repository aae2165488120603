"""The build-time fences of csrc/ (check_isa.sh runs them on every object's gfx950 disassembly): the scalar-base hazard scanner on a
synthetic listing - a vector-memory instruction reading a scalar register 0 / 5 wait states after the v_readfirstlane that wrote it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCANNER = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd", "csrc", "check_hazard.py")

LISTING = """
\tv_readfirstlane_b32 s5, v171                               // 000000001000: 7E0A05AB
\tv_readfirstlane_b32 s4, v170                               // 000000001004: 7E0805AA
%s\tglobal_store_dwordx2 v34, v[168:169], s[4:5]               // 000000001008: DC748000 0004A822
\ts_add_u32 s8, s8, s9                                       // 000000001010: 80080908
\tglobal_load_lds_dwordx4 v1, s[8:9]                         // 000000001014: DDF48000 00080001
"""


def run(listing):
    return subprocess.run([sys.executable, SCANNER, "synthetic.o"], input=listing, capture_output=True, text=True)


def test_scalar_base_hazard_is_flagged():
    r = run(LISTING % "")
    assert r.returncode == 1 and "reads s4" in r.stderr and "global_store_dwordx2" in r.stderr, r.stderr


def test_enough_wait_states_pass():
    r = run(LISTING % "\ts_nop 4                                                    // 000000001006: BF800004\n")
    assert r.returncode == 0 and r.stderr == "", r.stderr


def test_scalar_alu_written_base_passes():
    r = run("\ts_add_u32 s4, s4, s6                                       // 0: 80040604\n"
            "\tglobal_store_dword v1, v2, s[4:5]                          // 4: DC708000 00040201\n")
    assert r.returncode == 0, r.stderr


def test_middle_register_of_a_descriptor_is_flagged():
    """A v_readfirstlane into a MIDDLE register of a 4-SGPR buffer descriptor s[8:11] (the scanner once tested the two end registers only)."""
    r = run("\tv_readfirstlane_b32 s9, v3                                 // 0: 7E120503\n"
            "\tbuffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen lds        // 4: E05D1000 80020401\n")
    assert r.returncode == 1 and "reads s9" in r.stderr, r.stderr


def test_scan_stops_at_an_unconditional_branch_but_crosses_a_conditional_one():
    base = ("\tv_readfirstlane_b32 s4, v170                               // 0: 7E0805AA\n"
            "%s"
            "\tglobal_store_dword v1, v2, s[4:5]                          // 8: DC708000 00040201\n")
    assert run(base % "\ts_cbranch_scc0 12                                           // 4: BF84000C\n").returncode == 1
    assert run(base % "\ts_branch 12                                                 // 4: BF82000C\n").returncode == 0
