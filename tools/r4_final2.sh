set -x
mkdir -p gpurun_out/r4z
R=$(pwd); cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4z/prof_s1 -o f32w_s1 -- python3 $R/bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-io --modes '' > $R/gpurun_out/r4z/prof_s1_bench.json 2> $R/gpurun_out/r4z/prof_s1_err.log; cd $R
cut -c1-120 gpurun_out/r4z/prof_s1_bench.json
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r4z/gputests.log 2>&1; tail -12 gpurun_out/r4z/gputests.log
