set -x
mkdir -p gpurun_out/r4g
python bench.py --detail gpurun_out/r4g/detail.json > gpurun_out/r4g/bench_line.json 2> gpurun_out/r4g/bench_err.log && cut -c1-300 gpurun_out/r4g/bench_line.json
bash tools/pmc_traffic.sh gpurun_out/r4g/pmc f32w > gpurun_out/r4g/pmc.log 2>&1; tail -12 gpurun_out/r4g/pmc.log
