#!/bin/bash
# Artefacts behind the numbers of a round (run on the MI355X box from the repo root): default bench line + per-launch table, rocprofv3
# kernel stats of the same command (3 streams, and 1 stream for attribution), PMC traffic of the conv kernels (separate passes),
# per-layer tables of the Winograd kernels, SQ counters.      tools/final_profiles.sh <tag>
set -x
T=${1:-r6}
mkdir -p gpurun_out/$T
python bench.py --detail gpurun_out/$T/detail.json > gpurun_out/$T/bench_line.json 2> gpurun_out/$T/bench_err.log && cut -c1-200 gpurun_out/$T/bench_line.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -o f32w -- python3 $R/bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-io --no-clock-probes --modes '' > $R/gpurun_out/$T/prof_bench.json 2> $R/gpurun_out/$T/prof_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_s1 -o f32w_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-configs --no-cpu-baseline --no-io --no-clock-probes --modes '' > $R/gpurun_out/$T/prof_s1_bench.json 2> $R/gpurun_out/$T/prof_s1_err.log
cd $R
bash tools/pmc_traffic.sh gpurun_out/$T/pmc f32w > gpurun_out/$T/pmc.log 2>&1; tail -8 gpurun_out/$T/pmc.log
# (per-layer tables, SQ counters, the direct-form training line: tools/final_profiles_b.sh <tag>, a GPU call of its own)
ls gpurun_out/$T
