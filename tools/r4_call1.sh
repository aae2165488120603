set -x
mkdir -p gpurun_out/r4a
python bench.py > gpurun_out/r4a/bench_line.json 2> gpurun_out/r4a/bench_err.log && cut -c1-400 gpurun_out/r4a/bench_line.json
R=$(pwd); cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4a/prof -o f32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-io --modes '' > $R/gpurun_out/r4a/prof_bench.json 2> $R/gpurun_out/r4a/prof_err.log; cd $R
ls gpurun_out/r4a/prof | head
timeout -k 10 420 python tools/eager_conv_probe.py 7 > gpurun_out/r4a/eager_b7.txt 2>&1; tail -30 gpurun_out/r4a/eager_b7.txt
