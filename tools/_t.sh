R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r13t2/prof -o train -- python3 $R/bench.py --no-configs --mode train --precision f32w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers > $R/gpurun_out/r13t2_train_line.json 2> $R/gpurun_out/r13t2_err.log
