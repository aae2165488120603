#!/bin/bash
# SQ counters of the fp32 weight-gradient kernel per layer of the training step (tools/bench_wgrad.py 1; run on the MI355X box):
#   tools/pmc_wgrad.sh <outdir>
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wgrad}); mkdir -p $OUT
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_wgrad.py 1 > $OUT/sq.log 2>&1 || echo "pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAIT_ANY --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_wgrad.py 1 > $OUT/sq2.log 2>&1 || echo "pass2 failed"
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = defaultdict(lambda: defaultdict(float))
order = {}
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wgrad_mfma_kernel" not in k: continue
            did = int(r["Dispatch_Id"])
            rows[(sub, did)][r["Counter_Name"]] += float(r["Counter_Value"])
            rows[(sub, did)]["_grid"] = float(r.get("Grid_Size", 0) or 0)
for sub in ("sq", "sq2"):
    ids = sorted(d for s, d in rows if s == sub)
    print("==", sub, len(ids), "dispatches (3 per layer: 2 warm-up + 1 timed, fp32 kernel only)")
    for n, d in enumerate(ids):
        if n % 3 != 2: continue
        c = rows[(sub, d)]
        print("layer %2d grid %8d " % (n // 3, c["_grid"]) + " ".join("%s=%.3g" % (k.replace("SQ_", ""), v) for k, v in sorted(c.items()) if k != "_grid"))
PY
