# same-box A/B: default library vs tools/wabl_libssm_hip.so (built with WALTFLAGS), forced kinds $@
for rep in 1 2; do
for k in "$@"; do
  echo "== kind $k default"; NO_DIRECT=1 timeout -k 10 120 python tools/bench_layers_wino.py 7 736 1280 $k 2>&1 | grep -E "TOTAL" | cut -c60-110
  echo "== kind $k alt"; SSM_HIP_LIB=tools/wabl_libssm_hip.so NO_DIRECT=1 timeout -k 10 120 python tools/bench_layers_wino.py 7 736 1280 $k 2>&1 | grep -E "TOTAL" | cut -c60-110
done; done
