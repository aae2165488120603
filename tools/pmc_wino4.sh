#!/bin/bash
# SQ counters of the F(4x4,3x3) kernels per tile configuration on the per-layer bench (run on the MI355X box from the repo root):
#   tools/pmc_wino4.sh <outdir> [B]
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wino4}); mkdir -p $OUT
B=${2:-7}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
W4=1 NO_DIRECT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/sq.log 2>&1 || echo "pass failed"
W4=1 NO_DIRECT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/sq2.log 2>&1 || echo "pass2 failed"
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
            if "wino4" not in k or "pack" in k: continue
            m = re.search(r"(wino4p?_kernel)<.*?(W[48]Cfg<[^>]*>), (true|false)(?:, (true|false))?>", k.replace("(anonymous namespace)::", ""))
            cfg = "%s %s%s%s" % (m.group(1), m.group(2), " ups" if m.group(3) == "true" else "", " shuffle (sub-pixel interior)" if m.group(4) == "true" else "") if m else k[:60]
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for cfg, d in sorted(agg.items()):
        mf = max(d.get("SQ_INSTS_MFMA", 0.0), 1.0)
        lines = [cfg, "   vector instructions per MFMA %.2f, scalar %.2f, LDS %.2f, vector-memory %.3f" % (
            d.get("SQ_INSTS_VALU", 0) / mf - 1.0, d.get("SQ_INSTS_SALU", 0) / mf, d.get("SQ_INSTS_LDS", 0) / mf, d.get("SQ_INSTS_VMEM", 0) / mf),
            "   (SQ_INSTS_VALU counts the MFMAs too; 36 MFMAs of 32 cycles per wave and chunk)"]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("   LDS bank conflict cycles / LDS active cycles = %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]))
        wc = d.get("SQ_WAVE_CYCLES", 0)
        if wc:
            lines.append("   wave cycles waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) = %.3f" % (d.get("SQ_WAIT_INST_ANY", 0) / wc))
        for line in lines:
            print(line); fo.write(line + "\n")
PY
