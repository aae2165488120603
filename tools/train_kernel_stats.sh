#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training step (BASELINE config 3, mode f32w): tools/train_kernel_stats.sh <outdir> [steps]
# (the bench child runs --no-configs and starts no processes: the profiler's tool library has initialised the GPU before Python starts)
OUT=$(realpath ${1:-gpurun_out/train_stats}); mkdir -p $OUT
STEPS=${2:-10}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o train -- python3 $REPO/bench.py --no-configs --mode train --precision f32w --steps $STEPS --warmup 4 --no-cpu-baseline > $OUT/bench_line.json 2> $OUT/err.log
cd $REPO
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" $STEPS <<'PY'
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 4 + 5 + 3      # timed + warm-up + the 5 host-cost steps + the 3 bracketed ones
fam = defaultdict(lambda: [0.0, 0])
for r in rows:
    n = r["Name"]
    key = next((k for k in ("wgradw_kernel", "wgradw_finish", "wgrad_mfma_kernel", "wino4_kernel", "wino2_kernel", "wino7s_kernel", "wino5s_kernel",
                            "conv_mfma_kernel", "lrelu_bwd", "upsample_cat_bwd", "upsample2x_cat_kernel", "pack32", "splitk_finish", "synth_bwd", "inputs_bwd",
                            "maxpool", "sqdiff", "loss_terms", "multi_tensor", "FillFunctor", "copy_view", "elementwise", "avgpool") if k in n), n[:50])
    fam[key][0] += float(r["TotalDurationNs"]) / 1e6
    fam[key][1] += int(r["Calls"])
tot = sum(v[0] for v in fam.values())
print("kernel time per step (all streams summed; %d steps in the trace): %.2f ms" % (steps, tot / steps))
for k, (ms, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]:
    print("  %-26s %7.3f ms/step  %6.1f launches/step" % (k, ms / steps, calls / steps))
PY
