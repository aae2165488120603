#!/bin/bash
# rocprofv3 --kernel-trace --stats of the 4K line in mode f16 (BASELINE config 5 as worded): tools/kernel_stats_4k_f16.sh <outdir>
OUT=$(realpath ${1:-gpurun_out/stats_4k_f16}); mkdir -p $OUT
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o f16 -- python3 $REPO/bench.py --no-configs --size 4k --precision f16 --streams 1 --pairs-per-batch 1 --pairs-per-step 1 --steps 3 --warmup 1 --no-kernel-timers --no-clock-probes --no-cpu-baseline --no-io > $OUT/bench_line.json 2> $OUT/err.log
cd $REPO
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("  %6.2f %%  %7.3f ms avg  %5s calls  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e6, r["Calls"], r["Name"][:150]))
PY
