# share of the epilogue / DMA / barrier in the final form-2 kernel (diagnostics build)
export SSM_HIP_LIB=tools/wabl_libssm_hip.so
for a in 0 2 1 17; do
  echo "== SSM_WINO_ABL=$a"; NO_DIRECT=1 SSM_WINO_ABL=$a timeout -k 10 120 python tools/bench_layers_wino.py 7 2>&1 | grep -E "^conv|^fuse|TOTAL" | cut -c1-60,74-110
done
