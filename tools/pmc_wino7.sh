#!/bin/bash
# SQ counters of the blocked 7x7 kernel on the per-layer bench (run on the MI355X box from the repo root):
#   tools/pmc_wino7.sh <outdir> [B]
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wino7}); mkdir -p $OUT
B=${2:-14}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_layers_wino7.py $B > $OUT/sq.log 2>&1 || echo "pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_layers_wino7.py $B > $OUT/sq2.log 2>&1 || echo "pass2 failed"
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wino7_kernel" not in k and "wino7s_kernel" not in k: continue
            m = re.search(r"(wino7s?_kernel)<.*?(W7Cfg<[^>]*>)", k.replace("(anonymous namespace)::", ""))
            cfg = "%s %s" % (m.group(1), m.group(2)) if m else k[:60]
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for cfg, d in sorted(agg.items()):
        mf = max(d.get("SQ_INSTS_MFMA", 0.0), 1.0)
        lines = [cfg, "   vector instructions per MFMA %.2f, scalar %.2f, LDS %.2f, vector-memory %.3f" % (
            d.get("SQ_INSTS_VALU", 0) / mf - 1.0, d.get("SQ_INSTS_SALU", 0) / mf, d.get("SQ_INSTS_LDS", 0) / mf, d.get("SQ_INSTS_VMEM", 0) / mf),
            "   (SQ_INSTS_VALU counts the MFMAs too; 49 MFMAs of 32 cycles per wave and input channel; the F(2,7) kernel of the same bench run is not counted)"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("   LDS bank conflict cycles / LDS active cycles = %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]))
        wc = d.get("SQ_WAVE_CYCLES", 0)
        if wc:
            lines.append("   wave cycles waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) = %.3f" % (d.get("SQ_WAIT_INST_ANY", 0) / wc))
        for line in lines:
            print(line); fo.write(line + "\n")
PY
