#!/bin/bash
# SQ counters per conv16 configuration on the per-layer bench (run on the MI355X box from the repo root)
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_conv}); mkdir -p $OUT
BARGS=${2:-"7 0"}          # bench_layers16.py arguments: batch, fast (1: mode f16 = plain fp16, the config-5 path); size: $SSM_BENCH_H / $SSM_BENCH_W
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_layers16.py $BARGS > $OUT/sq.log 2>&1 || echo "pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_layers16.py $BARGS > $OUT/sq2.log 2>&1 || echo "pass2 failed"
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv16_kernel" not in k: continue
            cfg = k[k.index("Cfg16<"):k.index(">", k.index("Cfg16<")) + 1]
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
for cfg, d in agg.items():
    wc = d.get("SQ_WAVE_CYCLES", 1.0)
    print(cfg)
    for c in sorted(d):
        print("   %-28s %16.0f  %6.3f of WAVE_CYCLES" % (c, d[c], d[c] / wc))
PY
