set -x
mkdir -p gpurun_out
timeout 2400 python bench.py --steps 20 --warmup 3 --streams 3 --detail gpurun_out/r2k_detail.json > gpurun_out/r2k_bench_line.json 2> gpurun_out/r2k_bench_err.log
cut -c1-300 gpurun_out/r2k_bench_line.json; tail -n 3 gpurun_out/r2k_bench_err.log
bash tools/pmc_traffic.sh gpurun_out/r2k_pmc_f32 f32 > gpurun_out/r2k_pmc_f32.log 2>&1; tail -n 12 gpurun_out/r2k_pmc_f32.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2k_prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --pairs-per-step 4 --no-cpu-baseline --no-io --no-kernel-timers --modes '' --streams 1 > $GRAFT_REPO_ROOT/gpurun_out/r2k_prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/r2k_prof | head; find gpurun_out/r2k_prof -name "*kernel_stats.csv" | head -2
