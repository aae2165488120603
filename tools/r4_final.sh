set -x
mkdir -p gpurun_out/r4z
python bench.py --detail gpurun_out/r4z/detail.json > gpurun_out/r4z/bench_line.json 2> gpurun_out/r4z/bench_err.log && cut -c1-200 gpurun_out/r4z/bench_line.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z/prof -o f32w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-io --modes '' > $R/gpurun_out/r4z/prof_bench.json 2> $R/gpurun_out/r4z/prof_err.log; cd $R
bash tools/pmc_traffic.sh gpurun_out/r4z/pmc f32w > gpurun_out/r4z/pmc.log 2>&1; tail -8 gpurun_out/r4z/pmc.log
bash tools/pmc_wino.sh gpurun_out/r4z/pmc_wino 7 > gpurun_out/r4z/pmc_wino.log 2>&1; tail -3 gpurun_out/r4z/pmc_wino.log
