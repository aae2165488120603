set -e
timeout -k 10 300 python -m pytest tests/test_hip_wino7.py tests/test_hip_overshoot.py -x -q -m gpu > gpurun_out/r11x_w7s_tests.txt 2>&1 || { tail -30 gpurun_out/r11x_w7s_tests.txt; exit 1; }
tail -3 gpurun_out/r11x_w7s_tests.txt
for r in 1 2; do timeout -k 10 120 python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids | grep "conv1\|TOTAL"; done
