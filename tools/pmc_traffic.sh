#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes, no tracing domains mixed in):
#   tools/pmc_traffic.sh <outdir> [precision]      (run on the MI355X box, from the repo root; precision default f32)
set -e
OUT=$(realpath ${1:-gpurun_out/pmc}); mkdir -p $OUT
PREC=${2:-f32}
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
# 3 steps (1 warm-up + 2 timed) + the 3 steps of the host-enqueue measurement (r5), x 2 pairs, one stream, no side modes / CPU legs: 12 pairs through the kernels
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $REPO/bench.py --precision $PREC --steps 2 --warmup 1 --pairs-per-step 2 --pairs-per-batch 2 --no-configs --no-cpu-baseline --no-io --no-kernel-timers --modes '' --streams 1 > $OUT/$C.log 2>&1 || echo "pmc pass $C failed"
done
cd $REPO
python3 tools/pmc_summarize.py $OUT 12
