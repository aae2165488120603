set -x
mkdir -p gpurun_out
for cfg in "2 3" "2 2" "4 2" "1 3" "3 2" "4 1"; do set -- $cfg
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-io --no-kernel-timers --modes '' --pairs-per-step 12 --pairs-per-batch $1 --streams $2 > gpurun_out/r4q_pb$1_s$2.json 2>> gpurun_out/r4q_err.log
echo "pb=$1 streams=$2: $(cut -c1-90 gpurun_out/r4q_pb$1_s$2.json)"
done
