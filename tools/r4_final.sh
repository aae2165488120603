# final artefacts of the round: bench line + per-launch table, rocprofv3 kernel stats (3 streams and 1 stream), PMC traffic, SQ counters
set -x
T=${1:-r5a}
mkdir -p gpurun_out/$T
python bench.py --detail gpurun_out/$T/detail.json > gpurun_out/$T/bench_line.json 2> gpurun_out/$T/bench_err.log && cut -c1-200 gpurun_out/$T/bench_line.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof -o f32w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-io --modes '' > $R/gpurun_out/$T/prof_bench.json 2> $R/gpurun_out/$T/prof_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_s1 -o f32w_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-io --modes '' > $R/gpurun_out/$T/prof_s1_bench.json 2> $R/gpurun_out/$T/prof_s1_err.log
cd $R
bash tools/pmc_traffic.sh gpurun_out/$T/pmc f32w > gpurun_out/$T/pmc.log 2>&1; tail -6 gpurun_out/$T/pmc.log
bash tools/pmc_wino.sh gpurun_out/$T/pmc_wino 7 > gpurun_out/$T/pmc_wino.log 2>&1; tail -3 gpurun_out/$T/pmc_wino.log
timeout -k 10 200 python tools/bench_layers_wino.py 7 > gpurun_out/$T/layers_b7.txt 2>&1; tail -2 gpurun_out/$T/layers_b7.txt
