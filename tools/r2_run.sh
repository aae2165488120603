set -x
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_hip_dynamic_range.py tests/test_hip_backward.py tests/test_hip_recurrent.py tests/test_hip_model.py tests/test_hip_conv_f32.py -q -m gpu -s > gpurun_out/r2m_tests.log 2>&1; tail -n 30 gpurun_out/r2m_tests.log
