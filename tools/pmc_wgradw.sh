#!/bin/bash
# SQ counters of the Winograd-domain weight-gradient kernel per tile configuration on the per-layer bench (run on the MI355X box from the
# repo root; two separate --pmc passes, no tracing domains):   tools/pmc_wgradw.sh <outdir>
set -e
OUT=$(realpath ${1:-gpurun_out/pmc_wgradw}); mkdir -p $OUT
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
export WW_LAYERS=conv8a,conv10a,conv11a,conv11b SSM_WGRADW_TARGET=128
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/tools/bench_wgradw.py 5 > $OUT/sq.log 2>&1 || echo "pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -o pmc -- python3 $REPO/tools/bench_wgradw.py 5 > $OUT/sq2.log 2>&1 || echo "pass2 failed"
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
            if "wgradw_kernel" not in k and "wgrad_mfma_kernel" not in k: continue
            m = re.search(r"WwCfg<([^>]*)>", k)
            cfg = ("wgradw_kernel<WwCfg<%s>>" % m.group(1)) if m else "wgrad_mfma_kernel (direct form, all instantiations)"
            agg[cfg][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for cfg, d in sorted(agg.items()):
        mf = max(d.get("SQ_INSTS_MFMA", 0.0), 1.0)
        lines = [cfg, "   per MFMA (32x32x2, 64 cycles): vector %.2f, scalar %.2f, LDS %.2f, vector-memory %.3f" % (
            d.get("SQ_INSTS_VALU", 0) / mf - 1.0, d.get("SQ_INSTS_SALU", 0) / mf, d.get("SQ_INSTS_LDS", 0) / mf, d.get("SQ_INSTS_VMEM", 0) / mf)]
        if d.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"]))
        wc = d.get("SQ_WAVE_CYCLES", 0)
        if wc:
            lines.append("   wave cycles waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) = %.3f; matrix pipe busy / SQ busy = %.3f" % (
                d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(d.get("SQ_BUSY_CYCLES", 1.0), 1.0)))
        for line in lines:
            print(line); fo.write(line + "\n")
PY
