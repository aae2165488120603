#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (separate passes, no tracing domains mixed in):
#   tools/pmc_traffic.sh <outdir>      (run on the MI355X box, from the repo root)
set -e
OUT=$(realpath ${1:-gpurun_out/pmc}); mkdir -p $OUT
REPO=$(pwd); cd /tmp; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers --streams 1 > $OUT/$C.log 2>&1 || echo "pmc pass $C failed"
done
cd $REPO
python3 tools/pmc_summarize.py $OUT
