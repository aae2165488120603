#!/bin/bash
# rocprofv3 --kernel-trace --stats of the recurrent configuration (BASELINE config 4, mode f32w): tools/recurrent_kernel_stats.sh <outdir> [steps]
OUT=$(realpath ${1:-gpurun_out/recurrent_stats}); mkdir -p $OUT
STEPS=${2:-6}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o rec -- python3 $REPO/bench.py --no-configs --mode recurrent --precision f32w --steps $STEPS --warmup 2 --no-cpu-baseline --detail $OUT/detail.json > $OUT/bench_line.json 2> $OUT/err.log
cd $REPO
f=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" $STEPS $OUT/detail.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 2 + 3      # timed + warm-up + the 3 bracketed ones
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
print("kernel time per clip (%d clips in the trace): %.2f ms" % (steps, tot / steps))
for r in rows[:22]:
    print("  %-100s %7.3f ms/clip %6.1f launches/clip  avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[:100], float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3))
d = json.load(open(sys.argv[3]))
print("per-launch brackets (ms per clip, TFLOP/s direct-form):")
for n, v in sorted(d.items(), key=lambda kv: -kv[1]["ms_per_step"])[:40]:
    print("  %-40s %7.3f  %7.1f" % (n, v["ms_per_step"], v["tflops"]))
PY
