#!/bin/bash
# SQ counters per conv_mfma_kernel configuration (mode f32) on the per-layer bench (run on the MI355X box from the repo root)
#   tools/pmc_conv_f32.sh <outdir> [B]
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_conv_f32}); mkdir -p $OUT
B=${2:-7}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_layers.py $B > $OUT/sq.log 2>&1 || echo "pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_layers.py $B > $OUT/sq2.log 2>&1 || echo "pass2 failed"
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_mfma" not in k: continue
            cfg = k[k.index("Cfg<"):k.index(">", k.index("Cfg<")) + 1] if "Cfg<" in k else k[:60]
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for cfg, d in agg.items():
        wc = d.get("SQ_WAVE_CYCLES", 1.0)
        for line in [cfg] + ["   %-28s %16.0f  %6.3f of WAVE_CYCLES" % (c, d[c], d[c] / wc) for c in sorted(d)]:
            print(line); fo.write(line + "\n")
PY
