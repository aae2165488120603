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


def _strip_comments(src):
    import re
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def test_no_process_wide_guard_on_a_per_device_attribute():
    """hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to the (kernel, device) pair.  A std::call_once / function-static
    flag around it opts in the FIRST device only, and a second GPU driven from the same process (the reference's DataParallel
    replica threads, scripts/main.py:74-76) is refused at launch.  The one sanctioned caller is
    ssm::reserve_lds (ssm_common.h, bit per device) and conv16's (kernel, device) set."""
    import glob
    import re
    csrc = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp")))
    assert len(files) >= 12
    callers = {}
    for f in files:
        src = _strip_comments(open(f).read())
        for m in re.finditer(r"hipFuncSetAttribute\s*\(", src):
            callers.setdefault(os.path.basename(f), []).append(src[max(0, m.start() - 600):m.start()])
        # a call_once whose body (up to the closing "});") touches a per-device HIP call
        for m in re.finditer(r"call_once\s*\(", src):
            body = src[m.start():src.find("});", m.start()) + 3]
            assert not re.search(r"hipFuncSet|hipDeviceSet|hipSetDevice|hipFuncGetAttributes", body), (f, body[:200])
    assert sorted(callers) == ["ssm_common.h", "ssm_conv16.hip"], sorted(callers)
    for name, ctxs in callers.items():
        for ctx in ctxs:
            assert "hipGetDevice" in ctx, "%s: hipFuncSetAttribute without a device key in front of it" % name
    # every > 64 KiB launcher goes through the per-device helper with its own per-kernel mask
    for name in ("ssm_wino.hip", "ssm_wino1d.hip", "ssm_wino4.hip", "ssm_wino5.hip", "ssm_wino7.hip"):
        src = _strip_comments(open(os.path.join(csrc, name)).read())
        assert "ssm::reserve_lds(lds_reserved" in src and "static std::atomic<uint64_t> lds_reserved" in src, name


def test_header_states_the_device_contract_the_code_implements():
    hdr = open(os.path.join(ROOT, "include", "ssm_hip.h")).read()
    assert "takes the device from the pointers" not in hdr
    assert "CALLING THREAD's current device" in hdr and "keyed on (kernel, device)" in hdr
    csrc = os.path.join(ROOT, "superslomo-videointerpolation-pytorch_amd", "csrc")
    import glob
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cpp")):
        assert "hipSetDevice" not in _strip_comments(open(f).read()), f
