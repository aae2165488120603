#!/bin/bash
# Build-time fence for the packed-fp32 hazard: no v_pk_*_f32 instruction may appear in the gfx950 code of any
# object of libssm_hip.so.  (Vectoriser-generated v_pk_mul_f32 -> v_pk_add_f32 with swapped op_sel halves gave wrong lanes
# 48..63 in rare waves while another queue ran the block-scaled MFMA convolutions; every file is therefore compiled with
# -fno-slp-vectorize, and this check fails the build if a compiler or flag change brings packed fp32 back.)
#   usage: check_isa.sh a.o b.o ...
set -e
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
tmp=$(mktemp -d)
trap 'rm -rf $tmp' EXIT
bad=0
for o in "$@"; do
    objcopy -O binary --only-section=.hip_fatbin "$o" $tmp/fat.bin 2>/dev/null || continue
    [ -s $tmp/fat.bin ] || continue                     # host-only object
    $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
    $LLVM/llvm-objdump -d $tmp/dev.co > $tmp/dev.s
    python3 "$(dirname "$0")/check_hazard.py" "$o" < $tmp/dev.s || bad=1          # VALU-written SGPR -> vector-memory base (see the script)
    # scratch fence: the accumulator arrays of the kernels must live in registers (a loop that stays rolled - e.g. ssm_wino7.hip below its
    # -pragma-unroll-threshold - indexes them at run time and the compiler moves them to scratch memory: correct, and several times slower;
    # a register budget that is too tight spills).  Every object with device code is fenced; the allow-list names the kernels that may
    # touch scratch and why: the weight-repack kernels copy their job record / index small per-thread tables at run time - they run
    # once per plan (inference) or once per step over a few MB (training) and are HBM-bound.
    ALLOW='pack16q_kernel|pack32_batch_kernel'
    ns=$(awk -v allow="$ALLOW" '/^[0-9a-f]+ <.*>:/{name=$2} /scratch_/{ if (name !~ allow) c++ } END{print c+0}' $tmp/dev.s)
    if [ "$ns" != "0" ]; then
        echo "check_isa: $o contains $ns scratch-memory instructions outside the allow-listed repack kernels (an array left the registers, or a spill):" >&2
        awk -v allow="$ALLOW" '/^[0-9a-f]+ <.*>:/{name=$2} /scratch_/{ if (name !~ allow) c[name]++ } END{for (n in c) print "   ", c[n], n}' $tmp/dev.s >&2
        bad=1
    fi
    n=$(grep -c 'v_pk_[a-z0-9]*_f32' $tmp/dev.s || true)
    if [ "$n" != "0" ]; then
        echo "check_isa: $o contains $n packed-fp32 instructions (v_pk_*_f32):" >&2
        grep 'v_pk_[a-z0-9]*_f32' $tmp/dev.s | head -5 >&2
        bad=1
    fi
done
[ $bad = 0 ] && echo "check_isa: no packed-fp32 instructions, no scalar-base hazards, no scratch outside the repack kernels: $# object(s)"
exit $bad
