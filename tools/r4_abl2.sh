# ablation of the second Winograd form (diagnostics build): 1 = no LDS-DMA in the loop, 16 = no barrier, 2 = no epilogue, 4 = no B fetch / transform, 8 = no A fetch
export SSM_HIP_LIB=tools/wabl_libssm_hip.so
for a in 0 4 8 12 19 31; do
  echo "== SSM_WINO_ABL=$a"; NO_DIRECT=1 SSM_WINO_ABL=$a timeout -k 10 120 python tools/bench_layers_wino.py 7 2>&1 | grep -E "conv4b|conv9b|conv10b|fuse_conv|conv9a|conv11b|TOTAL" | cut -c1-60,74-110
done
