#!/bin/bash
# second half of tools/final_profiles.sh (per-layer tables, SQ counters, the direct-form training line) as its own GPU call
set -x
T=${1:-r6}
mkdir -p gpurun_out/$T
W4=1 timeout -k 10 200 python tools/bench_layers_wino.py 7 > gpurun_out/$T/wino4_layers_b7.txt 2>&1; tail -2 gpurun_out/$T/wino4_layers_b7.txt
timeout -k 10 200 python tools/bench_layers_wino.py 7 > gpurun_out/$T/wino2_layers_b7.txt 2>&1; tail -2 gpurun_out/$T/wino2_layers_b7.txt
timeout -k 10 200 python tools/bench_layers_wino7.py 14 > gpurun_out/$T/wino7_layers_b14.txt 2>&1; tail -2 gpurun_out/$T/wino7_layers_b14.txt
timeout -k 10 200 python tools/bench_layers_wino5.py 14 > gpurun_out/$T/wino5_layers_b14.txt 2>&1; tail -2 gpurun_out/$T/wino5_layers_b14.txt
bash tools/pmc_wino7.sh gpurun_out/$T/pmc_wino7 14 > gpurun_out/$T/pmc_wino7.log 2>&1; tail -6 gpurun_out/$T/pmc_wino7.log
bash tools/pmc_wino4.sh gpurun_out/$T/pmc_wino4 7 > gpurun_out/$T/pmc_wino4.log 2>&1; tail -30 gpurun_out/$T/pmc_wino4.log
python bench.py --mode train --no-cpu-baseline > gpurun_out/$T/train_f32_line.json 2>> gpurun_out/$T/bench_err.log
ls gpurun_out/$T
