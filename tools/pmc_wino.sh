#!/bin/bash
# SQ counters per Winograd kernel configuration on the per-layer bench (run on the MI355X box from the repo root)
#   tools/pmc_wino.sh <outdir> [B]
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wino}); mkdir -p $OUT
B=${2:-7}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
NO_DIRECT=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/sq.log 2>&1 || echo "pass failed"
NO_DIRECT=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_layers_wino.py $B > $OUT/sq2.log 2>&1 || echo "pass2 failed"
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
            if "wino" not in k or "pack" in k: continue
            cfg = ("wino2 " if "wino2" in k else "wino1 ") + k[k.index("WCfg<"):k.index(">", k.index("WCfg<")) + 1] + (" ups" if ", true>" in k else "")
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for cfg, d in agg.items():
        wc = d.get("SQ_WAVE_CYCLES", 1.0)
        for line in [cfg] + ["   %-28s %16.0f  %6.3f of WAVE_CYCLES" % (c, d[c], d[c] / wc) for c in sorted(d)]:
            print(line); fo.write(line + "\n")
PY
