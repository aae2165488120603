#!/bin/bash
# HBM bytes per launch of chosen F(4x4,3x3) layers (default: the 32-cout full-resolution layers conv11b / fuse_conv at batch 14), beside
# their durations - are they HBM-bound?  Separate passes: kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE (FETCH_SIZE x2: gfx950).
#   tools/pmc_wino4_hbm.sh <outdir> [B] [layers]          (run on the MI355X box from the repo root)
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wino4_hbm}); mkdir -p $OUT
B=${2:-14}
export ONLY=${3:-conv11b,fuse_conv} W4=1 NO_DIRECT=1
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/kt.log 2>&1 || echo "trace pass failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/$C.log 2>&1 || echo "pmc pass $C failed"
done
cd $REPO
python3 - $OUT $B <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, B = sys.argv[1], int(sys.argv[2])
print(open(os.path.join(out, "kt.log")).read())
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wino4_kernel" in r["Kernel_Name"]:
            dur[(r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
by = defaultdict(lambda: defaultdict(list))
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, C, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino4_kernel" in r["Kernel_Name"] and r["Counter_Name"] == C:
                by[(r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")))][C].append(float(r["Counter_Value"]))
for key in sorted(by):
    d = by[key]
    name = key[0].replace("(anonymous namespace)::", "")[:70]
    fe = 2.0 * sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024       # KB -> bytes, x2 (gfx950: 128-B requests tallied at 64 B)
    wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024
    us = sorted(dur.get(key, [0.0]))
    med = us[len(us) // 2]
    print("%s grid %s: %d launches, median %.1f us | read %.1f MB written %.1f MB per launch -> %.2f TB/s over the launch" % (
        name, key[1], len(us), med, fe / 1e6, wr / 1e6, (fe + wr) / 1e6 / max(med, 1e-9)))
PY
