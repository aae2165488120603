set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_conv_f32.py -x -q -m gpu > gpurun_out/r2s_tests.log 2>&1; tail -n 2 gpurun_out/r2s_tests.log
TUNE_ONLY=K7,K5,K3N32,K3N64,K3N64M,K3N64T,K3N128S,K3N128,K3N32GS,K3N64GS timeout 600 python tools/tune_conv_f32.py 14 2 > gpurun_out/r2s_tune_b14.log 2>&1; tail -n 1 gpurun_out/r2s_tune_b14.log
