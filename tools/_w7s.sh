set -e
timeout -k 10 300 python -m pytest tests/test_hip_wino7.py tests/test_hip_overshoot.py -x -q -m gpu > gpurun_out/r11v_w7s_tests.txt 2>&1 || { tail -30 gpurun_out/r11v_w7s_tests.txt; exit 1; }
tail -3 gpurun_out/r11v_w7s_tests.txt
echo "== phase probe, split kernel (single-frequency column pass over 7 waves), cin 32"; SSM_HIP_LIB=$PWD/tools/w7tr_libssm_hip.so timeout -k 10 120 python tools/wino7_phase_probe.py 32 14 2>&1 | grep -v amdgpu.ids
for r in 1 2; do
echo "== column pass in pairs on half 1 (first version)"; SSM_HIP_LIB=$PWD/tools/w7cp_libssm_hip.so timeout -k 10 120 python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids | grep "conv1\|TOTAL"
echo "== single-frequency column pass over 7 waves"; timeout -k 10 120 python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids | grep "conv1\|TOTAL"
done
