#!/bin/bash
# Compile-time ablation ladder of wgradw_kernel (csrc/ssm_wgradw.hip WW_ABL; libraries: `make wwalt WWTAG=a<n> WWFLAGS=-DWW_ABL=<n>`), on the
# three tile configurations' largest layers at 64 workgroups per launch (the loop, not the atomics): tools/ww_abl.sh <outfile>
OUT=${1:-gpurun_out/ww_abl.txt}
for v in "" a1 a2 a4 a8 a6 a7; do
  lib=""; [ -n "$v" ] && lib=tools/ww${v}_libssm_hip.so
  echo "=== WW_ABL ${v:-0 (shipped)}" >> $OUT
  SSM_HIP_LIB=$lib SSM_WGRADW_TARGET=64 WW_LAYERS=conv8a,conv10a,conv11a,conv11b python tools/bench_wgradw.py 20 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
