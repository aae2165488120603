set -x
mkdir -p gpurun_out
timeout 900 python bench.py --mode train --steps 20 --warmup 3 --detail gpurun_out/r2p_train_f32_detail.json > gpurun_out/r2p_train_f32.json 2> gpurun_out/r2p_err.log; cut -c1-250 gpurun_out/r2p_train_f32.json; tail -n 3 gpurun_out/r2p_err.log
timeout 900 python bench.py --mode train --steps 20 --warmup 3 --no-perceptual > gpurun_out/r2p_train_f32_noperc.json 2>> gpurun_out/r2p_err.log; cut -c1-250 gpurun_out/r2p_train_f32_noperc.json
timeout 1200 python -m pytest tests/test_hip_backward.py -q -m gpu -x > gpurun_out/r2p_tests.log 2>&1; tail -n 3 gpurun_out/r2p_tests.log
