# every Winograd tile configuration forced in turn on the 3x3 layer shapes (batch $1, default 7)
B=${1:-7}
for k in ${KINDS:-0 1 2 3 4 5 6 7 8 9 10 11 12 13 14}; do
  echo "== kind $k"; NO_DIRECT=1 timeout -k 10 120 python tools/bench_layers_wino.py $B 736 1280 $k 2>&1 | grep -E "^conv|^fuse|TOTAL|skipped" | cut -c1-60,74-110
done
