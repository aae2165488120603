# ablation sweep of the Winograd kernel (diagnostics build: make -C superslomo-videointerpolation-pytorch_amd/csrc wabl)
export SSM_HIP_LIB=tools/wabl_libssm_hip.so
for a in 0 1 17 2 4 8 12 13 31; do
  echo "== SSM_WINO_ABL=$a"; SSM_WINO_ABL=$a timeout -k 10 120 python tools/bench_layers_wino.py 7 2>&1 | grep -E "conv4b|conv9b|conv10b|fuse_conv|conv9a|conv11a|TOTAL" | cut -c1-125
done
