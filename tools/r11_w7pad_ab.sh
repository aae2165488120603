#!/bin/bash
# r11 (round 5): same-box A/B of the padded V layout of the blocked 7x7 form (csrc/ssm_wino7.hip W7_VPAD) + the wino4 workgroup timeline
# usage (GPU box): bash tools/r11_w7pad_ab.sh   (needs `make w7alt W7TAG=nopad W7FLAGS=-DW7_VPAD=0` and `make wtrace` in csrc)
set -o pipefail
O=gpurun_out
python -m pytest tests/test_hip_wino7.py -x -q -m gpu > $O/r11d_wino7_tests.txt 2>&1 || { tail -20 $O/r11d_wino7_tests.txt; exit 1; }
tail -2 $O/r11d_wino7_tests.txt
for round in 1 2; do
  echo "== round $round: padded (shipped)"; python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids
  echo "== round $round: r4 layout (W7_VPAD=0)"; SSM_HIP_LIB=tools/w7nopad_libssm_hip.so python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids
done > $O/r11d_wino7_vpad_ab.txt 2>&1
cat $O/r11d_wino7_vpad_ab.txt
for l in conv10b:1100 conv3b:1100 conv5b:300; do
  SSM_W4_TRACE_BLOCK=${l#*:} W4KIND=3 python tools/wino4_timeline.py ${l%:*} 2>&1 | grep -v amdgpu.ids | head -12
done > $O/r11d_wino4_timeline.txt 2>&1
cat $O/r11d_wino4_timeline.txt
