set -x
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_hip_ops.py tests/test_hip_backward.py tests/test_hip_recurrent.py tests/test_hip_model.py -q -m gpu -s -x > gpurun_out/r2n_tests.log 2>&1; tail -n 12 gpurun_out/r2n_tests.log
