set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_conv_f32.py tests/test_hip_ops.py -x -q -m gpu > gpurun_out/r2x_tests.log 2>&1; tail -n 2 gpurun_out/r2x_tests.log
for rep in 1 2; do
timeout 600 python tools/bench_layers.py 14 > gpurun_out/r2x_A$rep.log 2>&1; tail -n 1 gpurun_out/r2x_A$rep.log
SSM_HIP_LIB=$PWD/tools/alt_libssm_hip.so timeout 600 python tools/bench_layers.py 14 > gpurun_out/r2x_B$rep.log 2>&1; tail -n 1 gpurun_out/r2x_B$rep.log
done
timeout 600 python tools/tune_conv_f32.py 14 2 > gpurun_out/r2x_tune_b14.log 2>&1; tail -n 1 gpurun_out/r2x_tune_b14.log
