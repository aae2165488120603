#!/bin/bash
# ablation sweep of the blocked 7x7 form: every tools/w7*_libssm_hip.so (make w7alt W7TAG=.. W7FLAGS=..) through the layer bench
out=gpurun_out/${1:-w7_abl}.log
: > $out
for lib in lib/libssm_hip.so tools/w7*_libssm_hip.so; do
    [ "$lib" = lib/libssm_hip.so ] && lib=superslomo-videointerpolation-pytorch_amd/lib/libssm_hip.so
    echo "=== $lib" >> $out
    SSM_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/bench_layers_wino7.py 14 2>&1 | grep -v amdgpu.ids | tail -4 >> $out || exit 1
done
cat $out
